"""Seeded random scenarios (tests/scenarios_fuzz.py) on which the REAL reference was run (tests/golden/fuzz_reference.py keep ->
fuzz_l1.json, fuzz_l2.json, fuzz_l2_arrays.npz): the seeds kept from campaigns of 360 random databases x 2 samples x 4 walks, 950 whole command lines, 500 samples in random file shapes and
3 270 random layer-2 clusters in which the oracle, the product's host logic and (on the GPU) the HIP path agreed with the reference
on every seed -- once the reference's alpha grid took log10 / pow from libm as its pinned numpy 1.17.3 does (DESIGN.md section 4:
under numpy 1.26's own SIMD log10 / pow the reference itself flips between `no report` and `a strain at 1e-16` in ~1.5 % of
random clusters, the ones where the cross-validation picks the largest alpha; entries with "res_keys_numpy_1_26" are such seeds).

CPU: oracle + cst.Walk.  GPU: the product behind the reference's entry points."""
import importlib.util
import json
import os

import numpy as np
import pytest

from tests import scenarios_fuzz as sf

HERE = os.path.dirname(os.path.abspath(__file__))


def _module(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def fr():
    return _module(os.path.join(HERE, "golden", "fuzz_reference.py"), "fuzz_reference")


@pytest.fixture(scope="module")
def fp():
    return _module(os.path.join(os.path.dirname(HERE), "scripts", "r6", "fuzz_product.py"), "fuzz_product")


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def _arrs(golden_dir, seed):
    a = np.load(os.path.join(golden_dir, "fuzz_l2_arrays.npz"))
    d = {k.split("_", 1)[1]: a[k] for k in a.files if k.startswith("%d_" % seed)}
    return d or None


def test_kept_seeds_are_the_committed_ones(golden_dir):
    assert sorted(int(k) for k in _load(golden_dir, "fuzz_l1.json")) == sorted(sf.FUZZ_L1_KEPT)
    g2 = _load(golden_dir, "fuzz_l2.json")
    assert sorted(int(k) for k in g2) == sorted(sf.FUZZ_L2_KEPT)
    # what the campaign is for: several columns through ElasticNetCV, empty results, seeds on the alpha_max edge
    assert max(v.get("p") or 0 for v in g2.values()) >= 14
    assert any(v.get("p") and not v["res"] for v in g2.values())
    assert sum("res_keys_numpy_1_26" in v for v in g2.values()) >= 2
    assert sum(v["error"] == "IndexError" for v in g2.values()) >= 2          # np.percentile of [] (identify_strains_L2_Enet_Pscan_new_sp.py:114)


@pytest.mark.parametrize("seed", sf.FUZZ_L1_KEPT)
def test_fuzz_l1_oracle_and_host_walk(seed, golden_dir, fr, tmp_path):
    g = _load(golden_dir, "fuzz_l1.json")[str(seed)]
    assert fr.check_l1(g, str(tmp_path)) == []


@pytest.mark.parametrize("seed", sf.FUZZ_L1X_KEPT)
def test_fuzz_l1x_oracle_and_host_walk(seed, golden_dir, fr, tmp_path):
    """kmer.fa with rows that no node lists -- copies of node rows in front of and behind the original (the LAST row of a k-mer is the
    one jellyfish's dump is credited to, identify.py:91-95), rows with an N, lower-case rows (identify_low_mem.py and
    identify_low_depth.py look rows up as written: KeyError) -- through both modules and identify_ranks."""
    g = _load(golden_dir, "fuzz_l1x.json")[str(seed)]
    assert g["x"] and fr.check_l1(g, str(tmp_path)) == []


@pytest.mark.gpu
@pytest.mark.parametrize("seed", sf.FUZZ_L1X_KEPT)
def test_fuzz_l1x_hip_path(seed, golden_dir, fp, tmp_path, monkeypatch):
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    g = _load(golden_dir, "fuzz_l1x.json")[str(seed)]
    assert fp.product_l1(g, str(tmp_path)) == []


@pytest.mark.parametrize("seed", sf.FUZZ_L2_KEPT)
def test_fuzz_l2_oracle(seed, golden_dir, fr):
    g = _load(golden_dir, "fuzz_l2.json")[str(seed)]
    assert fr.check_l2(g, _arrs(golden_dir, seed)) == []


@pytest.mark.parametrize("seed", sf.FUZZ_FLOW_KEPT)
def test_fuzz_flow_oracle(seed, golden_dir, fr, tmp_path):
    """The reference's whole command line (flags, -k, paired / .gz input, Memory_DB, a cluster without its k-mer set: the reports
    written before the reference dies there) against the oracle's serial restatement."""
    g = _load(golden_dir, "fuzz_flow.json")[str(seed)]
    assert not sf.flow_known_deviation(seed, g["memory_db"])
    assert fr.check_flow(g, str(tmp_path)) == []


def test_flow_seeds_cover_what_they_are_kept_for(golden_dir):
    g = _load(golden_dir, "fuzz_flow.json")
    assert sorted(int(k) for k in g) == sorted(sf.FUZZ_FLOW_KEPT)
    assert {v["error"] for v in g.values()} >= {None, "FileNotFoundError"}
    dead = [v for v in g.values() if v["error"] == "FileNotFoundError"]
    assert any(len([f for f in v["files"] if f.endswith("StrainVote.report")]) >= 3 for v in dead)     # reports written before it died
    assert {int(v["argv"][v["argv"].index("-k") + 1]) for v in g.values() if "-k" in v["argv"]} >= {19, 21, 25, 27}
    assert sum(sf.flow_variant(int(k))["paired"] for k in g) >= 3 and sum(sf.flow_variant(int(k))["gz"] for k in g) >= 3
    assert any(v["memory_db"] for v in g.values())


@pytest.mark.parametrize("seed", sf.FUZZ_FMT_KEPT)
def test_fuzz_fmt_oracle(seed, golden_dir, fr, tmp_path):
    """The shapes a FASTA / FASTQ file comes in (wrapped records, CRLF, no final newline, '@' or '+' opening a quality line, '+name',
    blank lines at the end, lower case, two files of different formats, .gz) through the REAL jellyfish behind identify.py's zcat pipe:
    the oracle's reader and counter give the same counts (sha256 over the rows).  Three kept seeds are inputs on which the reference's
    pipeline LOSES reads (scenarios_fuzz.fmt_known_deviation): there the oracle, like the product, counts more -- asserted as such."""
    g = _load(golden_dir, "fuzz_fmt.json")[str(seed)]
    why = sf.fmt_known_deviation(g["kinds"])
    bad = fr.check_fmt(g, str(tmp_path))
    if why is None:
        assert bad == []
    else:
        assert len(bad) == 1 and bad[0][2] == "counts differ from jellyfish's" and bad[0][3] > bad[0][4], (why, bad)


def test_fmt_seeds_cover_every_shape(golden_dir):
    g = _load(golden_dir, "fuzz_fmt.json")
    kinds = {k.split("+")[0] for v in g.values() for k in v["kinds"]}
    assert kinds == set(sf.FMT_KINDS)
    assert {sf.fmt_known_deviation(v["kinds"]) for v in g.values()} == {None, "one of two files is .gz", "FASTQ without a final newline", "a .gz pair, FASTQ then FASTA"}


@pytest.mark.gpu
@pytest.mark.parametrize("seed", sf.FUZZ_FMT_KEPT)
def test_fuzz_fmt_hip_path(seed, golden_dir, fp, tmp_path, monkeypatch):
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    g = _load(golden_dir, "fuzz_fmt.json")[str(seed)]
    why = sf.fmt_known_deviation(g["kinds"])
    bad = fp.product_fmt(g, str(tmp_path))
    if why is None:
        assert bad == []
    else:                                                     # the product counts the reads the reference's pipeline loses
        assert len(bad) == 1 and bad[0][2] == "counts differ from jellyfish's" and bad[0][3] > bad[0][4], (why, bad)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", sf.FUZZ_FLOW_KEPT)
def test_fuzz_flow_hip_path(seed, golden_dir, fp, tmp_path, monkeypatch):
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    g = _load(golden_dir, "fuzz_flow.json")[str(seed)]
    assert fp.product_flow(g, str(tmp_path)) == []


@pytest.mark.gpu
@pytest.mark.parametrize("seed", sf.FUZZ_L1_KEPT)
def test_fuzz_l1_hip_path(seed, golden_dir, fp, tmp_path, monkeypatch):
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    g = _load(golden_dir, "fuzz_l1.json")[str(seed)]
    assert fp.product_l1(g, str(tmp_path)) == []


@pytest.mark.gpu
@pytest.mark.parametrize("seed", sf.FUZZ_L2_KEPT)
def test_fuzz_l2_hip_path(seed, golden_dir, fp):
    g = _load(golden_dir, "fuzz_l2.json")[str(seed)]
    assert fp.product_l2(g, _arrs(golden_dir, seed)) == []


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 5, 8])
def test_fuzz_flow_hip_path_sharded(world, golden_dir, tmp_path):
    """The kept command lines once more with the reads sharded over several rank processes on one device (gloo; rank 0 owns the output
    directory): layer 1's exchange of node statistics AND layer 2's all-reduced cluster tables, the `.gz` inputs inflated in range
    mode, the runs that die on a missing k-mer set (every rank must take the serial loop together: a RuntimeError on the ranks behind
    rank 0, db.rank0_first) -- the same files as the reference wrote.  The campaign ran 850 command lines this way with 3 ranks (550 of them also with 5 and 8); what it
    found was every rank creating the output directory at the same moment (FileExistsError on two of three)."""
    import subprocess
    import sys
    d = tmp_path / "flows"
    d.mkdir()
    for k, g in _load(golden_dir, "fuzz_flow.json").items():
        (d / ("flow_%s.json" % k)).write_text(json.dumps(g))
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "scripts", "r6", "fuzz_product.py"), str(d), "flow", str(world)],
                       env=dict(os.environ, SS_FUZZ_VERBOSE="120"), capture_output=True, text=True, timeout=900)
    tail = r.stdout[-1500:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    assert "fuzz_product flow (%d ranks): %d seeds, 0 with a disagreement" % (world, len(sf.FUZZ_FLOW_KEPT)) in r.stdout, tail
