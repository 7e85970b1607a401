#!/bin/bash
# micro-benchmark of in-launch exchanges + the new sharded-plan tests + candidate-set probe
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r03_s9}
O=$REPO/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 180 tools/micro/sync_probe 2>&1) > $O/sync_probe.log; echo "sync_probe rc=$?"; cat $O/sync_probe.log | cut -c1-200
(timeout 900 python -m pytest tests/test_multigpu_gpu.py tests/test_full_size_gpu.py tests/test_ndt_gpu.py -m gpu -q 2>&1 | tail -30) > $O/pytest.log; echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-300
(timeout 600 python tools/r03_cfg4_probe.py 2>&1 | tail -4) | tee $O/cfg4_default.log | cut -c1-400
