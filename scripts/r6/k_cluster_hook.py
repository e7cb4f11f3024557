import os, sys, json
sys.path.insert(0, os.getcwd())
sys.argv = ["k_cluster.py"]
from strainscan_amd import _lib
hook = int(os.environ.get("HOOK", "0"))
_lib.check(_lib.lib().ss_test_hook(4, hook), "hook")
import runpy
runpy.run_path("scripts/r6/k_cluster.py", run_name="__main__")
