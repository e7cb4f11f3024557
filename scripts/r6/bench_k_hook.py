#!/usr/bin/env python3
"""bench_k.py under ss_test_hook(4, HOOK): 2 = every k through the per-position kernel, 3 = every k through the run-queue kernel."""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from strainscan_amd import _lib
_lib.check(_lib.lib().ss_test_hook(4, int(os.environ.get("HOOK", "0"))), "ss_test_hook")
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_k.py"), run_name="__main__")
