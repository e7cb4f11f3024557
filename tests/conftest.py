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


@pytest.fixture(scope="session")
def mid_dbs(tmp_path_factory):
    """The configs[0]-shaped database of tests/scenarios_mid.py (53 clusters / 157 strains / 105 nodes), with and without
    the Memory_DB marker, and its samples' FASTQ files: {"DB_M": info, "DB_Mmem": info, "reads": {name: (path, bytes)}}."""
    from tests import scenarios_mid as sm
    root = str(tmp_path_factory.mktemp("ss_mid"))
    out = {"DB_M": sm.build_mid(root), "DB_Mmem": sm.build_mid(root, memory_db=True), "reads": {}}
    for sname in sm.mid_samples(out["DB_M"]):
        reads = sm.mid_reads(out["DB_M"], sname)
        p = os.path.join(root, sname + ".fq")
        with open(p, "wb") as f:
            f.write(reads)
        out["reads"][sname] = (p, reads)
    return out


@pytest.fixture(scope="session")
def built_db(tmp_path_factory):
    """The Tree_database the reference's own builder wrote (tests/golden/built_tree_db.tar.gz, tests/scenarios_built.py), unpacked;
    and its samples' FASTQ files: {"tdb": dir, "reads": {name: (path, bytes)}}."""
    from tests import scenarios_built as sb
    root = str(tmp_path_factory.mktemp("ss_built"))
    tdb = sb.unpack_tree_database(os.path.join(GOLDEN, "built_tree_db.tar.gz"), root)
    out = {"tdb": tdb, "reads": {}}
    for sname in sb.BUILT_SAMPLES:
        reads = sb.built_reads(sname)
        p = os.path.join(root, sname + ".fq")
        with open(p, "wb") as f:
            f.write(reads)
        out["reads"][sname] = (p, reads)
    return out
