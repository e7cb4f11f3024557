"""Layer-1 cluster identification -- drop-in for library/identify.py.

Public names and meanings follow the reference module: identify_cluster(fq_path, db_dir,
cutoff) (identify.py:402), jellyfish_count(fq_path, db_dir) (:73), match_node (:115),
piecewise (:130), del_outlier (:106), read_tree_structure (:15), get_node_label (:45).
The k-mer scan that the reference delegates to an external `jellyfish` process runs on the
MI355X (strainscan_amd/csrc/ss_scan.hip); the per-node reductions run there too
(ss_nodes.hip).  No CPU fallback exists.
"""
import os
import sys
import time

import numpy as np

from . import cst
from .db import tree_image
from .tree import read_tree_structure  # noqa: F401  (re-exported, identify.py:15)

_PARAMS = cst.Params(low_mem=False)
_UPPER_KEYS = True          # kmer_index_dict[...upper()] at identify.py:94


def _paths(fq_path):
    if isinstance(fq_path, str):
        return [p for p in fq_path.split(" ") if p]
    return [p for p in fq_path if p]


def jellyfish_count(fq_path, db_dir):
    """identify.py:73-103 -> match_results (row of kmer.fa -> count, zero counts included).
    Returns a read-only mapping view over device-produced arrays instead of a dict."""
    img = tree_image(db_dir, _UPPER_KEYS)
    img.scan(_paths(fq_path))
    return img.match_results()


def del_outlier(profile):
    """identify.py:106-112: drop values >= 100 * median."""
    cutoff = 100 * np.median(profile)
    return [v for v in profile if not (v >= cutoff)]


def match_node(match_results, db_dir, node_id, valid_kmers=None):
    """identify.py:115-127 -> (len(valid_kmer), k_profile).  Host form for API parity (the walk
    itself uses the all-nodes device reduction)."""
    with open(os.path.join(db_dir, "kmers", str(node_id)), "r") as f:
        d = set(map(int, f.readlines()[0].rstrip().split(" ")))
    valid = [k for k in d if k in match_results]
    prof = [match_results[k] for k in valid if match_results[k] > 0]
    if prof:
        prof = del_outlier(prof)
    return len(valid), prof


def piecewise(cov_cutoff, cov, label, k_profile):
    """identify.py:130-136."""
    return cst.piecewise(cov_cutoff, cov, label, np.mean(k_profile) if len(k_profile) else float("nan"))


def get_node_label(db_dir, tree):
    return cst.get_node_label(db_dir, tree, _PARAMS)


def _trace(*a):
    if not os.environ.get("STRAINSCAN_QUIET"):
        print(*a)


def _identify(fq_path, db_dir, cutoff, params, upper_keys):
    start = time.time()
    from .db import prefetch_reads
    pre = prefetch_reads(_paths(fq_path))         # the reads load while the database image does
    try:
        img = tree_image(db_dir, upper_keys)
    finally:
        if pre is not None:
            pre.join()
    if not img.is_external:
        img.scan(_paths(fq_path))
    walk = cst.Walk(cst.ImageProvider(img), db_dir, cutoff, params, out=_trace)
    res = walk.run()
    _trace("- The total running time of tree search is ", str(time.time() - start), " s\n")
    return res


def identify_cluster(fq_path, db_dir, cutoff):
    """identify.py:402-504.  fq_path = (fq1, fq2 or ''), cutoff = [cov, weighted cov, abundance];
    returns defaultdict{leaf id: {cls_ab, cls_per, cls_cov, cls_total_num, cls_covered_num,
    strain, s_ab}}."""
    return _identify(fq_path, db_dir, cutoff, _PARAMS, _UPPER_KEYS)
