#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r03_s11}
O=$REPO/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
for c in 3 5 7 8 3 5; do
  (LSR_GICP_BALL_CELLS=$c timeout 300 python tools/r02_gicp_probe.py 2>&1 | tail -1 | cut -c1-200) | sed "s/^/cells $c: /" | tee -a $O/gicp_cells.log
done
(timeout 900 python -m pytest tests/test_gicp_gpu.py -m gpu -q 2>&1 | tail -5) > $O/pytest.log; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300
(LSR_GICP_BALL_CELLS=8 timeout 900 python -m pytest tests/test_gicp_gpu.py -m gpu -q 2>&1 | tail -5) > $O/pytest8.log; echo "pytest rc=$?"; tail -3 $O/pytest8.log | cut -c1-300
