"""Layer-1 identification for memory-lean databases -- drop-in for library/identify_low_mem.py
(used when <DB>/Memory_DB exists, StrainScan.py:188-195).

Same walk as strainscan_amd.identify with the thresholds of the low-memory variant
(SURVEY.md 3.3): weak < 500, strong >= 1500, leaves can be weak, no qualified-parents fallback,
no early return when the root is below the abundance cutoff (the reference then raises
IndexError, identify_low_mem.py:230-234), and kmer.fa keys are NOT upper-cased
(identify_low_mem.py:81): a lower-case database k-mer without an upper-case twin raises
KeyError as at :88.
"""
from . import cst
from . import identify as _id
from .db import tree_image
from .identify import del_outlier, match_node, piecewise  # noqa: F401  (identify_low_mem.py:93-127: the same three functions)
from .tree import read_tree_structure  # noqa: F401

_PARAMS = cst.Params(low_mem=True)
_UPPER_KEYS = False


def jellyfish_count(fq_path, db_dir):
    img = tree_image(db_dir, _UPPER_KEYS)
    img.scan(_id._paths(fq_path))
    return img.match_results()


def get_node_label(db_dir, tree):
    return cst.get_node_label(db_dir, tree, _PARAMS)


def identify_cluster(fq_path, db_dir, cutoff):
    """identify_low_mem.py:386-470."""
    return _id._identify(fq_path, db_dir, cutoff, _PARAMS, _UPPER_KEYS)
