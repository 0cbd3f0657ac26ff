#!/bin/bash
# profiles/traffic.json for every launch shape bench.py can report: 100 M rows (one GPU) and 50 / 25 / 12.5 M rows (what one
# rank of a 2 / 4 / 8-GPU run scans per launch).  FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes, nothing else
# traced; the program follows `--` directly.  Run from the repo root on an MI355X box; the file is copied to
# gpurun_out/traffic/ (the box's profiles/ does not travel back) -- move it to profiles/ and stamp it locally with
# `python tools/make_traffic_json.py --stamp-git`.
set -eo pipefail
OUT=${1:-gpurun_out/collect}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
export PYTHONPATH=.
for ROWS in 100000000 50000000 25000000 12500000; do
    SUF="_$ROWS"; [ "$ROWS" = 100000000 ] && SUF=""
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/bench_fetch$SUF" -o bench -- \
        python3 bench.py --rows $ROWS --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_fetch$SUF.log" 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/bench_write$SUF" -o bench -- \
        python3 bench.py --rows $ROWS --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_write$SUF.log" 2>&1
    python3 tools/make_traffic_json.py "$OUT" $ROWS > "$OUT/traffic$SUF.log" 2>&1
    echo "traffic passes at $ROWS rows done"
done
mkdir -p gpurun_out/traffic && cp profiles/traffic.json gpurun_out/traffic/traffic.json
