#!/bin/bash
# Round 6, VERDICT r5 #3: TCC hits / misses and duration of attn_outproj_image (and of the QKV product feeding it) with the
# default launch order and with SSW_XCD_AFFINITY=1 (contiguous row tiles per XCD + the image's workgroup on that XCD).
# PMC pass and kernel trace are separate runs; the program follows `--` directly.   usage: tools/profile_attn_affinity.sh [B]
set -eo pipefail
B=${1:-200}
OUT=gpurun_out/attn_aff
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
export PYTHONPATH=.
for aff in 0 1; do
  if [ "$aff" = 1 ]; then export SSW_XCD_AFFINITY=1; else unset SSW_XCD_AFFINITY; fi
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$aff" -o t -- python3 tools/perf_clip_b200.py "$B" > "$OUT/trace_$aff.log" 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_$aff" -o t -- python3 tools/perf_clip_b200.py "$B" > "$OUT/pmc_$aff.log" 2>&1
  mkdir -p "$OUT/both_$aff" && cp -r "$OUT/trace_$aff" "$OUT/pmc_$aff" "$OUT/both_$aff/"
  echo "# B=$B SSW_XCD_AFFINITY=$aff"
  python3 tools/summarise_pmc.py "$OUT/both_$aff" attn_outproj_image
  python3 tools/summarise_pmc.py "$OUT/both_$aff" "gemm_gldsILi4E" | grep -v "^kernel," || true
  grep -h "ms" "$OUT/trace_$aff.log" | tail -1 || true
done
rm -rf "$OUT"/trace_* "$OUT"/pmc_* "$OUT"/both_*
