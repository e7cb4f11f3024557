"""Minimal stand-in for the third-party `treelib` package (absent from this image, no network).

Used ONLY by tests/golden/make_golden.py so that the reference's own identify*.py -- and, since round 6, its database builder
library/Build_tree.py (Tree.add_node / depth / children / siblings / parent / leaves) -- can be
imported and run in the build container to generate golden vectors.  It is written from
treelib's documented behaviour (module layout treelib/{tree,node}.py with 1.6.1's attribute names, insertion-ordered node dict, per-node successor list in
creation order, Node.__lt__ on tag, tag defaulting to the identifier); it is not part of the
product and is never imported by strainscan_amd/ or by the tests.
"""
from .node import Node  # noqa: F401
from .tree import Tree  # noqa: F401
