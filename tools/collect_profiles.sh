#!/bin/bash
# Regenerates the files under profiles/ on an MI355X box (run from the repo root; see profiles/README.md).
# PMC passes run on their own, never together with a trace domain; the program follows `--` directly.
set -eo pipefail
R=${1:-r02}
OUT=gpurun_out/collect
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
export PYTHONPATH=.

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -o bench -- \
    python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_trace.log" 2>&1
bash tools/collect_traffic.sh "$OUT"   # FETCH_SIZE / WRITE_SIZE passes at 100 / 50 / 25 / 12.5 M rows per launch -> profiles/traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/clip_trace" -o clip -- \
    python3 tools/perf_clip_b200.py 200 > "$OUT/clip_trace.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/clip_pmc" -o clip -- \
    python3 tools/perf_clip_b200.py 200 > "$OUT/clip_pmc.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/text_trace" -o text -- \
    python3 tools/perf_text.py > "$OUT/text_trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/text16_trace" -o text16 -- \
    python3 tools/perf_text_batch.py > "$OUT/text16_trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/knn_trace" -o knn -- \
    python3 tools/perf_knn.py 1560000 > "$OUT/knn_trace.log" 2>&1

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/loop_trace" -o loop -- \
    python3 tools/profile_loop.py knn_prop2 120000 > "$OUT/loop_trace.log" 2>&1
python3 tools/round_trace.py "$OUT/loop_trace" > "$OUT/knn_prop2_round_timeline.txt" 2>&1 || true

# tower GEMM shapes against hipBLASLt (lab build: ssw_debug_gemm), and the fused attention launch's in-kernel phase stamps
python3 tools/perf_gemm.py 15 14 --lib > "$OUT/gemm_ab.txt" 2> "$OUT/gemm_ab.err" || true
( echo "# f32 rows (default)"; SSW_AO_STAMPS=1 python3 tools/attn_out_stamps.py; echo "# bf16 rows"; SSW_AO_STAMPS=1 SSW_CLIP_BF16_STREAM=1 python3 tools/attn_out_stamps.py ) 2>/dev/null | grep -v amdgpu > "$OUT/attn_out_stamps.txt" || true

( time python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" ) 2> "$OUT/bench.time"   # after the PMC passes: its roofline.traffic reads the file just made
python3 tools/summarise_pmc.py "$OUT/bench_fetch" scan_scores > "$OUT/fetch_summary.csv"
python3 tools/summarise_pmc.py "$OUT/bench_write" scan_scores > "$OUT/write_summary.csv"
( echo "# mean per launch of scan_scores_kernel, rocprofv3 --pmc (two passes), bench.py --steps 3 --warmup 1"; cat "$OUT/fetch_summary.csv"; grep WRITE_SIZE "$OUT/write_summary.csv" ) > "profiles/${R}_bench_100M_pmc_fetch_write.csv"
python3 tools/summarise_pmc.py "$OUT/clip_pmc" "" > "profiles/${R}_clip_b200_pmc_summary.csv"
mkdir -p gpurun_out/traffic && cp profiles/traffic.json gpurun_out/traffic/traffic.json   # profiles/ is not merged back: copy out
cp "$OUT/bench.json" "profiles/${R}_bench_100M_output.json"          # the ONE stdout line
cp bench_detail.json "profiles/${R}_bench_100M_detail.json"           # everything the line leaves out
cp "$OUT/loop_trace/loop_kernel_stats.csv" "profiles/${R}_knn_prop2_session_1560k_kernel_stats.csv"
cp "$OUT/knn_prop2_round_timeline.txt" "profiles/${R}_knn_prop2_round_timeline.txt"
cp "$OUT/bench_trace/bench_kernel_stats.csv" "profiles/${R}_bench_100M_kernel_stats.csv"
cp "$OUT/clip_trace/clip_kernel_stats.csv" "profiles/${R}_clip_b200_kernel_stats.csv"
cp "$OUT/knn_trace/knn_kernel_stats.csv" "profiles/${R}_knn_1560k_kernel_stats.csv"
cp "$OUT/text_trace/text_kernel_stats.csv" "profiles/${R}_clip_text_1x8_kernel_stats.csv"
cp "$OUT/text16_trace/text16_kernel_stats.csv" "profiles/${R}_clip_text_16x77_kernel_stats.csv"
cp "$OUT/gemm_ab.txt" "profiles/${R}_gemm_ab.txt"
cp "$OUT/attn_out_stamps.txt" "profiles/${R}_attn_out_stamps.txt"
for f in profiles/${R}_*; do cp "$f" "gpurun_out/collect/$(basename "$f")"; done   # the box's profiles/ does not travel back
echo "summaries copied to gpurun_out/collect/: move them to profiles/ and run tools/make_traffic_json.py --stamp-git"
