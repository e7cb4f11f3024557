"""Node of the treelib stand-in (see __init__.py).  Attribute names follow treelib 1.6.1's node.py
(_identifier, _tag, expanded, _predecessor{tree id: parent id}, _successors{tree id: [child ids]}, data,
_initial_tree_id) so that a pickle written by either loads in the other: Build_tree.py:329 pickles a
treelib.Tree into tree.pkl for single-cluster databases and identify.py:19-21 loads it back."""
from collections import defaultdict


class Node(object):
    def __init__(self, tag=None, identifier=None, expanded=True, data=None):
        self._identifier = identifier
        self._tag = identifier if tag is None else tag
        self.expanded = expanded
        self._predecessor = {}
        self._successors = defaultdict(list)
        self.data = data
        self._initial_tree_id = None

    @property
    def identifier(self):
        return self._identifier

    @property
    def tag(self):
        return self._tag

    def is_leaf(self, tree_id=None):
        tid = self._initial_tree_id if tree_id is None else tree_id
        return len(self._successors[tid]) == 0

    def __lt__(self, other):
        return self.tag < other.tag

    def __repr__(self):
        return "Node(%r)" % (self.identifier,)
