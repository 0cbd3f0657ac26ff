"""CPU: bench.py's bookkeeping that needs no GPU -- the PMC traffic record is used only for the tree it was
measured on (kernel, variant, launch shape and source hash must match), otherwise the bench line says null."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(tmp_path, monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fake_root = tmp_path / "repo"
    (fake_root / "profiles").mkdir(parents=True)
    for rel in mod.SCAN_SOURCES:
        dst = fake_root / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        dst.write_bytes(open(os.path.join(ROOT, rel), "rb").read())
    monkeypatch.setattr(mod, "ROOT", str(fake_root))
    return mod, fake_root


def test_traffic_record_is_keyed_on_kernel_shape_and_source(bench):
    mod, root = bench
    rec = {"kernel": mod.SCAN_KERNEL, "kernel_source_sha256": mod.scan_source_sha256(), "git_head": "abc123",
           "shapes": {"1000": {"rows_per_launch": 1000, "hbm_bytes_per_launch": 2052000.0},
                      "125": {"rows_per_launch": 125, "hbm_bytes_per_launch": 257000.0}}}
    path = root / "profiles" / "traffic.json"
    path.write_text(json.dumps(rec))
    val, note = mod.measured_traffic(1000)
    assert val == 2052000.0 and "abc123" in note
    assert mod.measured_traffic(125)[0] == 257000.0                              # one rank's launch of an 8-GPU run
    val, note = mod.measured_traffic(999)
    assert val is None and "125, 1000" in note                                   # another launch shape
    path.write_text(json.dumps(dict(rec, kernel="scan_scores_kernel<1,1,nt>")))
    assert mod.measured_traffic(1000)[0] is None                                 # another variant
    path.write_text(json.dumps(rec))
    with open(root / mod.SCAN_SOURCES[0], "ab") as f:                             # the kernel source changed
        f.write(b"\n// edited\n")
    val, note = mod.measured_traffic(1000)
    assert val is None and "source changed" in note
    path.unlink()
    assert mod.measured_traffic(1000)[0] is None


def test_committed_traffic_record_matches_the_committed_kernel():
    """the records under profiles/ belong to the scan kernel in this tree; the one-GPU shape and the shapes one rank of a
    2 / 4 / 8-GPU run launches (roofline.traffic must not be null at N > 1) are all there"""
    spec = importlib.util.spec_from_file_location("bench_real", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for rows in (100_000_000, 50_000_000, 25_000_000, 12_500_000):
        val, note = mod.measured_traffic(rows)
        assert val is not None, note
        assert 0.98 < val / tj["shapes"][str(rows)]["algorithmic_bytes_per_launch"] < 1.10
