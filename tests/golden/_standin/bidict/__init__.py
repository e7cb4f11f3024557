"""Minimal stand-in for the third-party `bidict` package (absent from this image, no network).

Used ONLY by tests/golden/make_golden.py so that the reference's own database builder (library/Build_tree.py: its
hierarchy() and build_tree() keep `bidict.bidict` maps and read them backwards through `.inv`) can write a Tree_database
here.  What Build_tree.py uses and nothing more: construction empty, item assignment / deletion / lookup, iteration, len,
copy(), and `.inv[value]`.  Values are unique in every use there (that is bidict's contract: a duplicate value raises)."""


class _Inverse(object):
    def __init__(self, owner):
        self._o = owner

    def __getitem__(self, value):
        return self._o._back[value]

    def __contains__(self, value):
        return value in self._o._back

    def __len__(self):
        return len(self._o._back)


class bidict(dict):
    def __init__(self, *a, **kw):
        super().__init__()
        self._back = {}
        for k, v in dict(*a, **kw).items():
            self[k] = v

    @property
    def inv(self):
        return _Inverse(self)

    inverse = inv

    def __setitem__(self, key, value):
        if key in self:
            old = super().__getitem__(key)
            if old == value:
                return
            del self._back[old]
        if value in self._back:
            raise ValueError("value %r is already bound to key %r (bidict.ValueDuplicationError)" % (value, self._back[value]))
        super().__setitem__(key, value)
        self._back[value] = key

    def __delitem__(self, key):
        value = super().__getitem__(key)
        super().__delitem__(key)
        del self._back[value]

    def copy(self):
        c = bidict()
        for k, v in self.items():
            c[k] = v
        return c
