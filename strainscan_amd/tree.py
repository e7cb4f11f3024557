"""Cluster-search-tree container with the ordering rules the reference inherits from treelib.

The reference builds a `treelib.Tree` in library/identify.py:15-42 and then relies on three
orderings of that container: `all_nodes()` (creation order, root first), `children(id)`
(creation order of the children) and `leaves()` (creation order).  This module keeps exactly
those, plus `Node.__lt__` on the tag (used to break ties at identify.py:205).
"""


class Node:
    __slots__ = ("identifier", "tag", "data", "parent", "children")

    def __init__(self, identifier):
        self.identifier = identifier
        self.tag = identifier          # treelib: tag defaults to the identifier
        self.data = None
        self.parent = None
        self.children = []

    def is_leaf(self):
        return not self.children

    def __lt__(self, other):
        return self.tag < other.tag

    def __repr__(self):
        return "Node(%r)" % (self.identifier,)


class Tree:
    def __init__(self):
        self._nodes = {}               # insertion ordered
        self.root = None

    def create_node(self, identifier, parent=None):
        if identifier in self._nodes:
            raise ValueError("duplicated node id %r" % (identifier,))
        n = Node(identifier)
        if parent is None:
            if self.root is not None:
                raise ValueError("a tree takes one root merely")
            self.root = n
        else:
            p = self._nodes[parent]    # KeyError when the parent line comes later, like treelib
            n.parent = p
            p.children.append(n)
        self._nodes[identifier] = n
        return n

    def get_node(self, identifier):
        return self._nodes.get(identifier)

    def __getitem__(self, identifier):
        return self._nodes[identifier]

    def __len__(self):
        return len(self._nodes)

    def all_nodes(self):
        return list(self._nodes.values())

    def leaves(self):
        return [n for n in self._nodes.values() if not n.children]

    def parent(self, identifier):
        return self._nodes[identifier].parent

    def children(self, identifier):
        return list(self._nodes[identifier].children)

    def siblings(self, identifier):
        n = self._nodes[identifier]
        if n.parent is None:
            return []
        return [c for c in n.parent.children if c is not n]

    def is_ancestor(self, ancestor, grandchild):
        p = self._nodes[grandchild].parent
        while p is not None:
            if p.identifier == ancestor:
                return True
            p = p.parent
        return False

    def paths_to_leaves(self):
        out = []
        for leaf in self.leaves():
            path = []
            n = leaf
            while n is not None:
                path.append(n.identifier)
                n = n.parent
            out.append(path[::-1])
        return out


def read_tree_structure(db_dir):
    """library/identify.py:15-42.  Lines `id <TAB> parent|N <TAB> children|N [<TAB> strain]`;
    nodes are created root first: if the last line is not the root the file is rotated to start
    at the root line, otherwise it is read backwards.  A fourth field marks a single-strain
    cluster.  Returns (tree, GCF{node: strain})."""
    import os
    GCF = {}
    with open(os.path.join(db_dir, "tree_structure.txt"), "r") as f:
        lines = f.readlines()
    if len(lines) == 1:
        # identify.py:19-21 unpickles a treelib.Tree from tree.pkl here; treelib is a third-party
        # dependency of the reference that this package does not carry.
        raise NotImplementedError(
            "single-cluster database (%s/tree.pkl is a pickled treelib.Tree): not supported" % db_dir)
    tree = Tree()
    if lines[-1].split("\t")[1] != "N":
        i = 0
        for i in range(0, len(lines)):
            if lines[i].split("\t")[1] == "N":
                break
        order = lines[i:] + lines[:i]
    else:
        order = lines[::-1]
    for ln in order:
        t = ln.rstrip().split("\t")
        if t[1] == "N":
            tree.create_node(int(t[0]))
        else:
            tree.create_node(int(t[0]), parent=int(t[1]))
        if len(t) == 4:
            GCF[tree.get_node(int(t[0]))] = t[3]
    return tree, GCF
