"""Tree of the treelib stand-in (see __init__.py); attribute names as in treelib 1.6.1's tree.py
(_identifier, _nodes, root)."""
from .node import Node


class Tree(object):
    node_class = Node

    def __init__(self, identifier=None):
        self._identifier = identifier if identifier is not None else "standin-tree"
        self._nodes = {}
        self.root = None

    def add_node(self, node, parent=None):
        if node.identifier in self._nodes:
            raise ValueError("duplicated node id")
        if parent is None:
            if self.root is not None:
                raise ValueError("a tree takes one root merely")
            self.root = node.identifier
        else:
            if parent not in self._nodes:
                raise KeyError("parent %r not in tree" % (parent,))
            self._nodes[parent]._successors[self._identifier].append(node.identifier)
        node._predecessor[self._identifier] = parent
        if node._initial_tree_id is None:
            node._initial_tree_id = self._identifier
        self._nodes[node.identifier] = node

    def create_node(self, tag=None, identifier=None, parent=None, data=None):
        n = Node(tag, identifier, data=data)
        self.add_node(n, parent)
        return n

    def get_node(self, nid):
        return self._nodes.get(nid)

    def __getitem__(self, nid):
        return self._nodes[nid]

    def _succ(self, nid):
        return self._nodes[nid]._successors[self._identifier]

    def _pred(self, nid):
        return self._nodes[nid]._predecessor.get(self._identifier)

    def all_nodes(self):
        return list(self._nodes.values())

    def leaves(self):
        return [n for n in self._nodes.values() if not self._succ(n.identifier)]

    def parent(self, nid):
        p = self._pred(nid)
        return None if p is None else self._nodes[p]

    def children(self, nid):
        return [self._nodes[i] for i in self._succ(nid)]

    def siblings(self, nid):
        p = self._pred(nid)
        if p is None:
            return []
        return [self._nodes[i] for i in self._succ(p) if i != nid]

    def is_ancestor(self, ancestor, grandchild):
        p = self._pred(grandchild)
        while p is not None:
            if p == ancestor:
                return True
            p = self._pred(p)
        return False

    def depth(self, node=None):
        """Levels below the root of `node` (a Node or an identifier); Build_tree.py:86 calls tree.depth(node=<Node>)."""
        nid = node.identifier if isinstance(node, Node) else node
        d = 0
        p = self._pred(nid)
        while p is not None:
            d += 1
            p = self._pred(p)
        return d

    def paths_to_leaves(self):
        res = []
        for leaf in self.leaves():
            path = []
            nid = leaf.identifier
            while nid is not None:
                path.append(nid)
                nid = self._pred(nid)
            res.append(path[::-1])
        return res
