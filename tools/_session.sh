#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 900 python -m pytest tests/test_ndt_gpu.py tests/test_multigpu_gpu.py tests/test_full_size_gpu.py -m gpu -q -x 2>&1 | tail -6 | cut -c1-300
timeout 600 python tools/r03_cfg4_stage_c.py 2>&1 | tail -2
NC=8 timeout 600 python tools/r03_cfg4_stage_c.py 2>&1 | tail -2
