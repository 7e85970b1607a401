#!/bin/bash
# side-stream launches on a bounded number of workgroups + eager fitness for a share of 8 (A/B by environment)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
O=gpurun_out/s9.txt; : > $O
run() { echo "[$*]" >> $O; for F in 0 24; do env "$@" FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1 | cut -c1-330 >> $O; done; }
run LSR_SIDE_CAP=0
run LSR_SIDE_CAP=512 LSR_FIT_EAGER_MIN=1
run LSR_SIDE_CAP=256 LSR_FIT_EAGER_MIN=1
run LSR_SIDE_CAP=1024 LSR_FIT_EAGER_MIN=1
run LSR_SIDE_CAP=512 LSR_FIT_EAGER_MIN=2
run LSR_SIDE_CAP=512 LSR_FIT_EAGER_MIN=4
run LSR_SIDE_CAP=512 LSR_FIT_EAGER_MIN=12
run LSR_SIDE_CAP=0 LSR_NDT_CHAINS=3
run LSR_SIDE_CAP=512 LSR_FIT_EAGER_MIN=1 LSR_NDT_CHAINS=3
cat $O
timeout 600 python -m pytest -x -q -m gpu tests/test_loop_closure_gpu.py tests/test_nn_gpu.py tests/test_full_size_gpu.py -k "cfg4 or loop or fitness or nn" 2>&1 | tail -3
