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


def _strict_loads(line):
    def refuse(name):
        raise ValueError(f"non-finite constant {name} in the bench line")
    return json.loads(line, parse_constant=refuse)


@pytest.mark.parametrize("stored", ["r05_bench_100M_output.json", "r05_rehearsal_gloo_world4_12p5M_rows_per_rank.json"])
def test_the_stdout_line_is_short_strict_json_with_the_contract_fields(stored):
    """the driver keeps 8 000 characters of stdout and could not parse round 5's 20.8-KB line.  The line
    is built from the full result (which goes to bench_detail.json): < 6 000 characters, strict JSON, numbers only in
    `extras`, and it still carries the contract's fields + roofline + cpu_baseline (+ allgather_us at N > 1)."""
    spec = importlib.util.spec_from_file_location("bench_line", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    full = json.load(open(os.path.join(ROOT, "profiles", stored)))
    assert len(json.dumps(full)) > 2000          # the stored record is a full one
    line = mod.compact_line(full)
    assert "\n" not in line and len(line) < mod.LINE_LIMIT <= 6000
    assert line.isascii()
    d = _strict_loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "allgather_us", "extras"):
        assert k in d, k
    assert d["value"] == pytest.approx(full["value"], rel=1e-5) and d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    assert r["traffic"] is not None and r["traffic"] == pytest.approx(full["roofline"]["traffic"], rel=1e-5)
    assert "workload" in d["config"] and "model" not in d["config"]
    if full["n_gpus"] == 1:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and len(c["sample"]) <= 120
        assert d["extras"]["c5_knn_prop2_hip_iters_per_s"] > 0 and d["extras"]["c3_image_b200_ms"] > 0
    else:
        assert d["allgather_us"]["steps"] == full["steps"] and d["cpu_baseline"] is None
        assert any(k.endswith("_iters_per_s_all_gpus") for k in d["extras"])
    assert all(isinstance(v, (int, float)) and not isinstance(v, bool) for v in d["extras"].values()), "extras: numbers only"


def test_the_line_never_outgrows_the_limit_and_never_carries_non_finite_numbers():
    spec = importlib.util.spec_from_file_location("bench_line2", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_100M_output.json")))
    big = full["extras"]["feedback_loop"]
    for i in range(200):   # a run that grew many more loops: the line drops extras from the end instead of growing
        big["full_120000x13"][f"loop{i}"] = dict(big["full_120000x13"]["plain"])
    full["roofline"]["avg_launch_ms"] = float("nan")
    full["extras"]["clip"]["image_ms_per_batch"] = float("inf")
    line = mod.compact_line(full)
    assert len(line) < mod.LINE_LIMIT
    d = _strict_loads(line)
    assert d["roofline"]["avg_launch_ms"] is None and "c3_image_b200_ms" not in d["extras"]
    assert d["roofline"]["frac"] is not None and d["cpu_baseline"]["value"] > 0


@pytest.mark.parametrize("stored", ["r06_rehearsal_gloo_world4_12p5M_rows_per_rank.json", "r06_rehearsal_gloo_world4_25M_rows_per_rank.json"])
def test_the_n_gt_1_line_as_a_run_printed_it_parses(stored):
    """a world-4 rehearsal line of the compact format, kept under profiles/: what a driver would read at N > 1"""
    line = open(os.path.join(ROOT, "profiles", stored)).read().strip()
    assert "\n" not in line and len(line) < 6000
    d = _strict_loads(line)
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["cpu_baseline"] is None
    assert d["roofline"]["traffic"] is not None and d["roofline"]["frac"] == pytest.approx(d["roofline"]["achieved"] / d["roofline"]["peak"], rel=1e-3)
    assert d["allgather_us"]["steps"] == d["steps"] and d["allgather_us"]["bytes_per_rank"] > 0
    assert d["config"]["rows_per_gpu"] * 4 == d["config"]["rows_total"]
    assert all(isinstance(v, (int, float)) and not isinstance(v, bool) for v in d["extras"].values())
