#!/bin/bash
cd /root/repo
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 900 python -m pytest tests/test_gicp_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -k "gicp or covar or variant" 2>&1 | tail -4
timeout 120 python tools/r02_gicp_probe.py 2>&1 | grep GICP
