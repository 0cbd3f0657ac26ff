#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a ONE-GPU box (SSW_BENCH_REHEARSAL=gloo): every rank uses GPU 0, the exchange goes
# through host tensors over gloo -- the same shards, kernels, messages and merge as the real run, only the collective's
# transport differs (RCCL cannot put two ranks on one device).  The line's NUMBERS MEAN NOTHING (N ranks share one GPU);
# what it shows is that the line is complete: config.rccl_ranks / config.collective, allgather_us, roofline.traffic for the
# rank's launch shape, the replicas of the feedback loop.  World size 4 is the most a one-GPU box admits here (the
# pool's process guard allows 6 processes on a card; 8 ranks are refused), so the 8-rank launch shape (12.5 M rows)
# is run as 4 ranks x 12.5 M rows = 50 M rows in total, and the 4-rank shape (25 M rows) as 4 x 25 M.
# Usage: bash tools/rehearse_multi_gpu.sh [out dir]; copies the two JSON lines to <out dir>/rehearsal_*.json
set -eo pipefail
OUT=${1:-gpurun_out/rehearsal}
mkdir -p "$OUT"
export SSW_BENCH_REHEARSAL=gloo PYTHONPATH=. HSA_ENABLE_IPC_MODE_LEGACY=0
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29531 \
    bench.py --gpus 4 --steps 10 --warmup 3 --rows 50e6 --loop-images 2000 > "$OUT/rehearsal_world4_12p5M_rows_per_rank.json" 2> "$OUT/rehearsal_a.err"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29532 \
    bench.py --gpus 4 --steps 10 --warmup 3 --rows 100e6 --no-extras > "$OUT/rehearsal_world4_25M_rows_per_rank.json" 2> "$OUT/rehearsal_b.err"
echo "rehearsal lines written to $OUT"
