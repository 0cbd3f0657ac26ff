#!/bin/bash
# Regenerates the files under profiles/ on an MI355X box (run from the repo root; see profiles/README.md).
# PMC passes run on their own, never together with a trace domain; the program follows `--` directly.
set -eo pipefail
R=${1:-r01}
OUT=gpurun_out/collect
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
export PYTHONPATH=.

python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -o bench -- \
    python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/bench_fetch" -o bench -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/bench_write" -o bench -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_write.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/clip_trace" -o clip -- \
    python3 tools/perf_clip_b200.py 200 > "$OUT/clip_trace.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/clip_pmc" -o clip -- \
    python3 tools/perf_clip_b200.py 200 > "$OUT/clip_pmc.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/knn_trace" -o knn -- \
    python3 tools/perf_knn.py 1560000 > "$OUT/knn_trace.log" 2>&1

cp "$OUT/bench.json" "profiles/${R}_bench_100M_output.json"
cp "$OUT/bench_trace/bench_kernel_stats.csv" "profiles/${R}_bench_100M_kernel_stats.csv"
cp "$OUT/clip_trace/clip_kernel_stats.csv" "profiles/${R}_clip_b200_kernel_stats.csv"
cp "$OUT/knn_trace/knn_kernel_stats.csv" "profiles/${R}_knn_1560k_kernel_stats.csv"
echo "raw PMC collections are under $OUT (bench_fetch, bench_write, clip_pmc): summarise as profiles/README.md describes"
