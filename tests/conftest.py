import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def l1_dbs(tmp_path_factory):
    """Synthetic Tree_database dirs of tests/scenarios.py, built once per session."""
    from tests import scenarios as sc
    root = str(tmp_path_factory.mktemp("ss_dbs"))
    return {name: sc.build_l1(name, root) for name in sc.L1_DBS}


@pytest.fixture(scope="session")
def l1_reads(l1_dbs, tmp_path_factory):
    """sample name -> (fastq path, bytes)"""
    from tests import scenarios as sc
    root = str(tmp_path_factory.mktemp("ss_reads"))
    out = {}
    for sname, (dbn, _, _) in sc.L1_SAMPLES.items():
        reads = sc.sample_reads(l1_dbs[dbn], sname)
        p = os.path.join(root, sname + ".fq")
        with open(p, "wb") as f:
            f.write(reads)
        out[sname] = (p, reads)
    return out
