#!/bin/bash
# PMC passes (own runs, no trace domain) of one tower-GEMM shape on two kernel variants: L2 hit / miss / requests and
# the SQ wave-cycle split.  usage: tools/profile_gemm_pmc.sh OUTDIR "M N K epi" variantA variantB
set -eo pipefail
OUT=${1:-gpurun_out/gemm_pmc}; SHAPE=${2:-"10000 3072 768 2"}; VA=${3:-14}; VB=${4:-21}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
for V in $VA $VB; do
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/tcc_v$V" -o g -- \
      python3 tools/perf_gemm_one.py $SHAPE $V 10 > "$OUT/tcc_v$V.log" 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
      --output-format csv -d "$OUT/sq_v$V" -o g -- python3 tools/perf_gemm_one.py $SHAPE $V 10 > "$OUT/sq_v$V.log" 2>&1
  python3 tools/summarise_pmc.py "$OUT/tcc_v$V" gemm > "$OUT/tcc_v$V.csv" || true
  python3 tools/summarise_pmc.py "$OUT/sq_v$V" gemm > "$OUT/sq_v$V.csv" || true
done
tail -n +1 "$OUT"/*.csv
