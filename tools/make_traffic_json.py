"""profiles/traffic.json from the raw rocprofv3 PMC passes of bench.py (tools/collect_profiles.sh):
    python tools/make_traffic_json.py <collect dir> [rows]     on the GPU box, after the FETCH_SIZE / WRITE_SIZE passes
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
    fetch, nf = mean_counter(os.path.join(root, "bench_fetch"), "FETCH_SIZE", "scan_scores_kernel")
    write, nw = mean_counter(os.path.join(root, "bench_write"), "WRITE_SIZE", "scan_scores_kernel")
    tj = {"kernel": bench.SCAN_KERNEL, "rows_per_launch": rows, "kernel_source_sha256": bench.scan_source_sha256(),
          "kernel_sources": list(bench.SCAN_SOURCES), "git_head": "unstamped",
          "fetch_size_kib_raw": fetch, "write_size_kib_raw": write, "launches_averaged": [nf, nw],
          "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads are tallied at 64 B per 128-B request), WRITE_SIZE x1, KiB->B x1024",
          "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024, "algorithmic_bytes_per_launch": rows * bench.ROW_BYTES}
    json.dump(tj, open(OUT, "w"), indent=1)
    print(json.dumps(tj, indent=1))


if __name__ == "__main__":
    main()
