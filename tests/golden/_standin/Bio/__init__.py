"""Minimal stand-in for Biopython (absent from this image, no network): ONLY `Bio.SeqIO.parse(path_or_handle, "fasta")`, which is
all the reference's library/Build_tree.py asks of it (extract_kmers: `str(seq_record.seq)` of every record).  Used only by
tests/golden/make_golden.py to let the reference's own builder write a Tree_database."""
