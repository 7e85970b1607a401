#!/bin/bash
# N1 with one-launch ingest + device-side grid dimensions: parity tests, timing A/B, kernel stats
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
O=gpurun_out/s10.txt; : > $O
timeout 900 python -m pytest -x -q -m gpu tests/test_voxelgrid_gpu.py tests/test_frontend_stream_gpu.py tests/test_loop_closure_gpu.py 2>&1 | tail -5 >> $O
for v in 1 0; do LSR_VG_DEVICE_DIMS=$v timeout 300 python tools/preprocess_probe.py 2>&1 | tail -1 >> $O; done
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_n1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_n1 -o n1 -- python $OLDPWD/tools/preprocess_probe.py > /dev/null 2>&1)
python tools/stats_to_md.py /tmp/prof_n1/n1_kernel_stats.csv 20 >> $O
cat $O
