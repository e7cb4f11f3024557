"""Low-depth cluster ranking (`-b 1`) -- drop-in for library/identify_low_depth.py.

identify_ranks(fq_path, db_dir) -> [(leaf id, score), ...] sorted by score, descending
(identify_low_depth.py:104-156).  Every node with at least 1000 valid k-mers contributes its
covered fraction; a leaf's score is the geometric mean over its root path of
x = 1 if frac > 0.05 else log10(180 * frac + 1).  The reference scans the reads twice
(:119,124) with identical results; one device scan is done here.
"""
import time
from math import log

from . import identify as _id
from .db import tree_image
from .tree import read_tree_structure

_UPPER_KEYS = False         # identify_low_depth.py:64 keys the dict by the raw text
MIN_VALID = 1000            # identify_low_depth.py:90-91


def Log(x):
    """identify_low_depth.py:97-101 (math.log(x, 10), not log10)."""
    x = 180 * x
    x = x + 1
    return log(x, 10)


def rank_paths(tree, frac):
    """identify_low_depth.py:134-151: frac = {node id: covered fraction or -1}."""
    tmp = {}
    for path in tree.paths_to_leaves():
        leaf = path[-1]
        score = 1
        N = len([i for i in path if frac[i] != -1])
        for i in path:
            if frac[i] == -1:
                continue
            x = 1 if frac[i] > 0.05 else Log(frac[i])
            score = score * pow(x, 1 / N)
        if score != 0:
            tmp[leaf] = score
    return sorted(tmp.items(), key=lambda kv: kv[1], reverse=True)


def identify_ranks(fq_path, db_dir):
    start = time.time()
    tree, _ = read_tree_structure(db_dir)
    from .db import prefetch_reads
    pre = prefetch_reads(_id._paths(fq_path))     # the reads load while the database image does
    try:
        img = tree_image(db_dir, _UPPER_KEYS)
    finally:
        if pre is not None:
            pre.join()
    if not img.is_external:
        img.scan(_id._paths(fq_path))
    st = img.node_stats()
    frac = {}
    for node in tree.all_nodes():
        s = st[img.node_index[node.identifier]]
        length = int(s["length"])
        if length < MIN_VALID:          # match_node returns (0, []) -> cov = -1 (:90-91,128-130)
            frac[node.identifier] = -1
        else:
            frac[node.identifier] = int(s["n_kept"]) / length
    ranked = rank_paths(tree, frac)
    _id._trace(ranked)
    _id._trace("- The total running time of the low-depth strain detection is ", str(time.time() - start), " s\n")
    return ranked
