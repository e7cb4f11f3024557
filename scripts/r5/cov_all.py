#!/usr/bin/env python3
"""Which statements of the package's Python files does the whole test suite (-m gpu and not) never execute?
   cov_all.py [pytest args ...]   -> gpurun_out/cov_all.txt   (exploration: what the per-file gates do not cover yet)"""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import pytest
from tests import covgate
files = sorted(glob.glob(os.path.join(ROOT, "strainscan_amd", "*.py")))
tr = covgate.LineTrace(*files)
with tr:
    rc = pytest.main(sys.argv[1:] or ["tests", "-q", "-x", "-p", "no:cacheprovider"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "cov_all.txt"), "w") as o:
    o.write("pytest rc %s\n" % rc)
    for f in files:
        ex = covgate.executable_lines(f)
        miss = covgate.unvisited(tr, f)
        o.write("== %s: %d of %d statements never ran\n" % (os.path.relpath(f, ROOT), len(miss), len(ex)))
        for ln, text in miss:
            o.write("   %4d  %s\n" % (ln, text))
print(open(os.path.join(ROOT, "gpurun_out", "cov_all.txt")).read()[:200])
