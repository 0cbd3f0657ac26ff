#!/bin/bash
# rocprofv3 evidence for the label-propagation sweep (run from the repo root on an MI355X box):
#   kernel trace + three PMC passes (each on its own, never with a trace domain; program directly after `--`)
# usage: tools/profile_labelprop.sh <tag> [nodes] [graph] [order]   -> gpurun_out/lp_<tag>/...   (graph / order: tools/perf_labelprop.py)
set -eo pipefail
TAG=${1:-run}
N=${2:-1560000}
G=${3:-random}
O=${4:-none}
OUT=gpurun_out/lp_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
export PYTHONPATH=.
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o lp -- python3 tools/perf_labelprop.py "$N" 200 "$G" "$O" > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o lp -- python3 tools/perf_labelprop.py "$N" 20 "$G" "$O" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o lp -- python3 tools/perf_labelprop.py "$N" 20 "$G" "$O" > "$OUT/write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/tcc" -o lp -- python3 tools/perf_labelprop.py "$N" 20 "$G" "$O" > "$OUT/tcc.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -o lp -- python3 tools/perf_labelprop.py "$N" 20 "$G" "$O" > "$OUT/sq.log" 2>&1
python3 tools/summarise_pmc.py "$OUT" k_lp > "$OUT/summary.csv"
cat "$OUT/summary.csv"
grep -h "per sweep" "$OUT/trace.log" || true
