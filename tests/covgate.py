"""Statement-coverage gate for the host mirrors of the reference's sequential logic.

The walk (strainscan_amd/cst.py) and the layer-2 drivers restate reference control flow statement by statement; a branch
no golden scenario enters is a branch nobody has compared with the reference.  `LineTrace` records the lines of chosen
source files that execute inside a `with` block (sys.settrace: no dependency); `executable_lines` lists what could have
(the line starts of every code object compiled from the file, minus docstrings and the `def` / `class` headers executed
at import); the gate tests fail on anything unvisited outside their allow-lists."""
import dis
import sys
import threading
import types


class LineTrace:
    def __init__(self, *files):
        self.files = set(files)
        self.hit = {f: set() for f in files}

    def _global(self, frame, event, arg):
        f = frame.f_code.co_filename
        if f not in self.files:
            return None
        hit = self.hit[f]
        hit.add(frame.f_lineno)

        def local(frame, event, arg):
            if event == "line":
                hit.add(frame.f_lineno)
            return local
        return local

    def __enter__(self):
        self._old = sys.gettrace()
        threading.settrace(self._global)          # the layer-2 batch runs clusters on worker threads
        sys.settrace(self._global)
        return self

    def __exit__(self, *exc):
        sys.settrace(self._old)
        threading.settrace(None)
        return False


def _walk(code, acc, top):
    """Lines of the statements inside function bodies (nested functions, lambdas and comprehensions included)."""
    for c in code.co_consts:
        if isinstance(c, types.CodeType):
            inner = set(ln for _, ln in dis.findlinestarts(c) if ln)
            if c.co_name not in ("<module>",) and not _is_class_body(c):
                inner.discard(c.co_firstlineno)          # the `def` line itself runs at import / class creation
                acc |= inner
            _walk(c, acc, False)


def _is_class_body(code):
    return "__qualname__" in code.co_names and "__module__" in code.co_names


def executable_lines(path):
    """-> sorted line numbers of the statements inside the functions and methods of the file."""
    with open(path) as f:
        src = f.read()
    acc = set()
    _walk(compile(src, path, "exec"), acc, True)
    lines = src.split("\n")
    out = []
    for ln in sorted(acc):
        t = lines[ln - 1].strip()
        if t.startswith(('"""', "'''", "def ", "class ", "@")) or t in ("", ")", "]", "}"):
            continue
        out.append(ln)
    return out


def unvisited(trace, path, allow=()):
    """Executable lines of `path` that `trace` never saw, minus the allow-list (line TEXT fragments, so that the list
    survives edits elsewhere in the file).  -> [(line number, text)]"""
    with open(path) as f:
        lines = f.read().split("\n")
    miss = []
    for ln in executable_lines(path):
        if ln in trace.hit[path]:
            continue
        text = lines[ln - 1].strip()
        if any(a in text for a in allow):
            continue
        miss.append((ln, text))
    return miss
