"""Minimal stand-in for the third-party `treelib` package (absent from this image, no network).

Used ONLY by tests/golden/make_golden.py so that the reference's own identify*.py can be
imported and run in the build container to generate golden vectors.  It is written from
treelib's documented behaviour (insertion-ordered node dict, per-node successor list in
creation order, Node.__lt__ on tag, tag defaulting to the identifier); it is not part of the
product and is never imported by strainscan_amd/ or by the tests.
"""


class Node(object):
    def __init__(self, tag=None, identifier=None, data=None):
        self.identifier = identifier
        self.tag = identifier if tag is None else tag
        self.data = data
        self._pred = None
        self._succ = []

    def is_leaf(self):
        return len(self._succ) == 0

    def __lt__(self, other):
        return self.tag < other.tag

    def __repr__(self):
        return "Node(%r)" % (self.identifier,)


class Tree(object):
    def __init__(self):
        self._nodes = {}
        self.root = None

    def create_node(self, tag=None, identifier=None, parent=None, data=None):
        n = Node(tag, identifier, data)
        if identifier in self._nodes:
            raise ValueError("duplicated node id")
        if parent is None:
            if self.root is not None:
                raise ValueError("a tree takes one root merely")
            self.root = identifier
        else:
            if parent not in self._nodes:
                raise KeyError("parent %r not in tree" % (parent,))
            self._nodes[parent]._succ.append(identifier)
            n._pred = parent
        self._nodes[identifier] = n
        return n

    def get_node(self, nid):
        return self._nodes.get(nid)

    def __getitem__(self, nid):
        return self._nodes[nid]

    def all_nodes(self):
        return list(self._nodes.values())

    def leaves(self):
        return [n for n in self._nodes.values() if n.is_leaf()]

    def parent(self, nid):
        p = self._nodes[nid]._pred
        return None if p is None else self._nodes[p]

    def children(self, nid):
        return [self._nodes[i] for i in self._nodes[nid]._succ]

    def siblings(self, nid):
        p = self._nodes[nid]._pred
        if p is None:
            return []
        return [self._nodes[i] for i in self._nodes[p]._succ if i != nid]

    def is_ancestor(self, ancestor, grandchild):
        p = self._nodes[grandchild]._pred
        while p is not None:
            if p == ancestor:
                return True
            p = self._nodes[p]._pred
        return False

    def paths_to_leaves(self):
        res = []
        for leaf in self.leaves():
            path = []
            nid = leaf.identifier
            while nid is not None:
                path.append(nid)
                nid = self._nodes[nid]._pred
            res.append(path[::-1])
        return res
