cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
python -m pytest tests/test_clip_gpu.py -x -q -m gpu 2>&1 | tail -3
SSW_AO_STAMPS=1 python tools/_tmp_stamps.py 2>&1 | grep -v amdgpu.ids
python3 tools/perf_clip_b200.py 200 2>&1 | grep -v amdgpu.ids
SSW_CLIP_UNFUSED_ATTN=1 python3 tools/perf_clip_b200.py 200 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4e/t -o clip -- python3 tools/perf_clip_b200.py 200 > gpurun_out/r4e/t.log 2>&1
grep -E "attn_outproj" gpurun_out/r4e/t/clip_kernel_stats.csv | cut -c1-200
