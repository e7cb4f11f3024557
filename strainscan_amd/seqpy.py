"""Drop-in for the reference's CPython extension library/seqpy.c: revcomp(str) -> str.

The IUPAC reverse complement runs in the native library (ss_revcomp, strainscan_amd/csrc/ss_host.hip;
a device form ss_revcomp_dev exists for batches of equal-length sequences).  Only the database
BUILD modules of the reference import seqpy; it is kept because BASELINE.json names it."""
from . import _lib


def revcomp(seq):
    return _lib.revcomp(seq)
