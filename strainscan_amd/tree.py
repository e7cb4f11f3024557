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
        # single-cluster database (Build_tree.py:283-334): identify.py:19-21 unpickles a treelib.Tree from
        # tree.pkl and returns it with an empty GCF
        pkl = os.path.join(db_dir, "tree.pkl")
        if os.path.exists(pkl):
            return load_tree_pkl(pkl), GCF
        tree = Tree()                     # the file the builder pickled holds exactly this node (Build_tree.py:284-286)
        tree.create_node(int(lines[0].split("\t")[0]))
        return tree, GCF
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


class _PickledObject:
    """Stands in for treelib.tree.Tree / treelib.node.Node while tree.pkl is read: keeps the pickled
    attribute dict, nothing else."""

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2:            # (dict state, slots state)
            state = dict(state[0] or {}, **(state[1] or {}))
        self.state = dict(state)


class _PickledTree(_PickledObject):
    pass


class _PickledNode(_PickledObject):
    pass


def load_plain_pkl(path):
    """id2strain_re.pkl (a list of strain names, Recls_withR_new.py:114-115) and all_kid.pkl (k-mer -> id dict,
    Build_kmer_sets..._sp.py:409-410) hold only builtin containers, strings and numbers: such a pickle names no
    global at all, so the Unpickler admits none -- a database file cannot make the identification call anything."""
    import pickle

    class _NoGlobals(pickle.Unpickler):
        def find_class(self, module, name):
            raise pickle.UnpicklingError("%s: %s.%s is not data (only lists, dicts, strings and numbers are read)"
                                         % (path, module, name))

    with open(path, "rb") as f:
        return _NoGlobals(f).load()


def load_tree_pkl(path):
    """tree.pkl of a single-cluster database: `pkl.dump(tree, ...)` of a treelib.Tree (Build_tree.py:329),
    read back by identify.py:19-21.  treelib is a dependency of the reference that this package does not
    carry, so the pickle is read with a restricted Unpickler -- only treelib's Tree and Node classes
    (mapped onto attribute bags) and collections.defaultdict / OrderedDict are admitted, nothing else in the
    file can name a callable -- and turned into this module's Tree: nodes in the pickled dict's order
    (= creation order), children in the order of the successor lists.  Understands the attribute layout of
    treelib >= 1.6 (_predecessor / _successors keyed by tree id; the reference pins 1.6.1) and of treelib
    <= 1.5 (_bpointer / _fpointer)."""
    import collections
    import pickle

    class _Restricted(pickle.Unpickler):
        def find_class(self, module, name):
            if module.split(".")[0] == "treelib" and name == "Tree":
                return _PickledTree
            if module.split(".")[0] == "treelib" and name == "Node":
                return _PickledNode
            if module == "collections" and name in ("defaultdict", "OrderedDict"):
                return getattr(collections, name)
            if module in ("builtins", "__builtin__") and name in ("list", "dict", "set", "int", "str", "object"):
                return {"list": list, "dict": dict, "set": set, "int": int, "str": str, "object": object}[name]
            if module in ("copy_reg", "copyreg") and name == "_reconstructor":
                import copyreg
                return copyreg._reconstructor
            raise pickle.UnpicklingError("tree.pkl: %s.%s is not part of a pickled treelib.Tree" % (module, name))

    with open(path, "rb") as f:
        obj = _Restricted(f).load()
    if not isinstance(obj, _PickledTree) or "_nodes" not in obj.state:
        raise ValueError("%s does not hold a treelib.Tree" % path)
    tid = obj.state.get("_identifier")
    raw = obj.state["_nodes"]

    def links(st):
        if "_successors" in st:                                     # treelib >= 1.6
            succ = st["_successors"]
            pred = st.get("_predecessor", {})
            ch = succ.get(tid) if tid in succ else (list(succ.values())[0] if len(succ) == 1 else [])
            par = pred.get(tid) if tid in pred else (list(pred.values())[0] if len(pred) == 1 else None)
            return par, list(ch or [])
        return st.get("_bpointer"), list(st.get("_fpointer") or [])  # treelib <= 1.5

    info = {}
    for nid, n in raw.items():
        if not isinstance(n, _PickledNode):
            raise ValueError("%s: node %r is not a treelib.Node" % (path, nid))
        info[nid] = links(n.state)
    tree = Tree()
    root = obj.state.get("root")
    order = list(raw)
    if root in info and order and order[0] != root:                  # parents are created before their children
        order.remove(root)
        order.insert(0, root)
    pending = list(order)
    while pending:
        rest = []
        for nid in pending:
            par = info[nid][0]
            if nid == root or par is None:
                tree.create_node(nid)
            elif tree.get_node(par) is not None:
                tree.create_node(nid, parent=par)
            else:
                rest.append(nid)
        if len(rest) == len(pending):
            raise ValueError("%s: nodes without a path to the root: %r" % (path, rest))
        pending = rest
    for nid, (_, ch) in info.items():                                # children in the successor lists' order
        node = tree[nid]
        by_id = {c.identifier: c for c in node.children}
        if sorted(by_id, key=repr) == sorted(ch, key=repr):
            node.children = [by_id[c] for c in ch]
    return tree
