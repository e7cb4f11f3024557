"""`SeqIO.parse(source, "fasta")` of the Biopython stand-in (see __init__.py): records with .id, .description and .seq (a str),
sequence lines joined as Biopython joins them."""


class _Record(object):
    def __init__(self, header, seq):
        self.description = header
        self.id = header.split()[0] if header.split() else ""
        self.name = self.id
        self.seq = seq


def parse(source, fmt):
    if fmt != "fasta":
        raise ValueError("the stand-in reads FASTA only")
    handle = open(source) if isinstance(source, str) else source
    try:
        header, chunks = None, []
        for line in handle:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if header is not None:
                    yield _Record(header, "".join(chunks))
                header, chunks = line[1:], []
            elif header is not None:
                chunks.append(line.strip())
        if header is not None:
            yield _Record(header, "".join(chunks))
    finally:
        if isinstance(source, str):
            handle.close()
