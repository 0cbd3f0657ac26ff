"""profiles/traffic.json from the raw rocprofv3 PMC passes of bench.py (tools/collect_profiles.sh):
    python tools/make_traffic_json.py <collect dir> [rows]     on the GPU box, after the FETCH_SIZE / WRITE_SIZE passes of that
                                                               launch shape (100 M rows: one GPU; 50 / 25 / 12.5 M: what one
                                                               rank of a 2 / 4 / 8-GPU run scans); records accumulate per shape
    python tools/make_traffic_json.py --stamp-git               locally: record the commit the measured tree came from
HBM bytes per launch are formed exactly as guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
FETCH_SIZE (KiB) x 1024 x 2 (wide coalesced reads are tallied at 64 B per 128-B request) + WRITE_SIZE (KiB) x 1024."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "profiles", "traffic.json")


def mean_counter(root, counter, kernel_filter):
    vals = []
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and kernel_filter in row["Kernel_Name"]:
                    vals.append(float(row["Counter_Value"]))
    assert vals, f"no {counter} rows for {kernel_filter} under {root}"
    return sum(vals) / len(vals), len(vals)


def main():
    if sys.argv[1] == "--stamp-git":
        tj = json.load(open(OUT))
        tj["git_head"] = subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=ROOT, text=True).strip()
        json.dump(tj, open(OUT, "w"), indent=1)
        print("stamped", tj["git_head"])
        return
    import bench
    root = sys.argv[1]
    rows = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
    # the passes of launch shape `rows` live under <root>/bench_fetch[_<rows>] and bench_write[_<rows>]
    suffix = "" if rows == 100_000_000 else f"_{rows}"
    fetch, nf = mean_counter(os.path.join(root, "bench_fetch" + suffix), "FETCH_SIZE", "scan_scores_kernel")
    write, nw = mean_counter(os.path.join(root, "bench_write" + suffix), "WRITE_SIZE", "scan_scores_kernel")
    sha = bench.scan_source_sha256()
    try:  # one record per launch shape, all of the same kernel source
        tj = json.load(open(OUT))
        if tj.get("kernel_source_sha256") != sha or tj.get("kernel") != bench.SCAN_KERNEL:
            tj = {}
    except Exception:
        tj = {}
    tj.update({"kernel": bench.SCAN_KERNEL, "kernel_source_sha256": sha, "kernel_sources": list(bench.SCAN_SOURCES),
               "git_head": "unstamped",
               "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads are tallied at 64 B per 128-B request), WRITE_SIZE x1, KiB->B x1024"})
    tj.setdefault("shapes", {})[str(rows)] = {
        "rows_per_launch": rows, "fetch_size_kib_raw": fetch, "write_size_kib_raw": write, "launches_averaged": [nf, nw],
        "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024, "algorithmic_bytes_per_launch": rows * bench.ROW_BYTES,
        "ratio_to_algorithmic": (fetch * 1024 * 2 + write * 1024) / (rows * bench.ROW_BYTES)}
    json.dump(tj, open(OUT, "w"), indent=1)
    print(json.dumps(tj, indent=1))


if __name__ == "__main__":
    main()
