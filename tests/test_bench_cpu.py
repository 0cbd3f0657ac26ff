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
    rec = {"kernel": mod.SCAN_KERNEL, "rows_per_launch": 1000, "kernel_source_sha256": mod.scan_source_sha256(),
           "hbm_bytes_per_launch": 2052000.0, "git_head": "abc123"}
    path = root / "profiles" / "traffic.json"
    path.write_text(json.dumps(rec))
    val, note = mod.measured_traffic(1000)
    assert val == 2052000.0 and "abc123" in note
    assert mod.measured_traffic(999)[0] is None                                  # another launch shape
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
    """the record under profiles/ belongs to the scan kernel in this tree"""
    spec = importlib.util.spec_from_file_location("bench_real", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    val, note = mod.measured_traffic(tj["rows_per_launch"])
    assert val is not None, note
    assert 0.98 < val / tj["algorithmic_bytes_per_launch"] < 1.10
