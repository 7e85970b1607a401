#!/bin/bash
# NN kernels: refine keeps its points in registers, group16 scans with two loads in flight — parity + cfg4 / share / gicp timings
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
O=gpurun_out/s12.txt; : > $O
timeout 1200 python -m pytest -x -q -m gpu tests/test_nn_gpu.py tests/test_gicp_gpu.py tests/test_loop_closure_gpu.py tests/test_full_size_gpu.py 2>&1 | tail -4 >> $O
(timeout 300 python tools/cfg4_stage_probe.py 2>&1 | tail -2; NC=8 timeout 300 python tools/cfg4_stage_probe.py 2>&1 | tail -2) >> $O
for F in 0 24; do FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1 | cut -c1-330 >> $O; done
timeout 300 python tools/gicp_probe.py 2>&1 | tail -3 >> $O
bash tools/kstats.sh nn tools/cfg4_probe.py 2>&1 | grep -i "nn_\|nn1\|==" >> $O
cat $O
