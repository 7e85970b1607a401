#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r03_s19
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
for w in 1 0 1 0; do
(LSR_NN_PREFETCH=$w timeout 600 python tools/r03_cfg4_probe.py 2>&1 | tail -3 | cut -c1-200) | sed "s/^/prefetch $w: /" | tee -a $O/cfg4.log
done
(timeout 900 python -m pytest tests/test_ndt_gpu.py tests/test_full_size_gpu.py tests/test_nn_gpu.py tests/test_loop_closure_gpu.py tests/test_multigpu_gpu.py -m gpu -q -x 2>&1 | tail -8) > $O/pytest.log; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-400
