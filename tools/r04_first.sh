#!/bin/bash
# round 4, first GPU session: the rewritten derivative kernels (canonical sums, lane kernel) — parity tests + cfg-4 stage timings
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r04_first
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 900 python -m pytest tests/test_ndt_gpu.py -q -x 2>&1 | tail -30) > $O/pytest_ndt.log; tail -15 $O/pytest_ndt.log | cut -c1-400
(timeout 600 python tools/r03_cfg4_probe.py 2>&1 | tail -5) | tee $O/cfg4_default.log
(LSR_NDT_WORKGROUP=512 timeout 600 python tools/r03_cfg4_probe.py 2>&1 | tail -3) | tee $O/cfg4_wg512.log
(LSR_NDT_WIDEN=0 timeout 600 python tools/r03_cfg4_probe.py 2>&1 | tail -3) | tee $O/cfg4_nowiden.log
