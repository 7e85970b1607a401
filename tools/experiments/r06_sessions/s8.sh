#!/bin/bash
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 900 python -m pytest tests/test_ndt_gpu.py tests/test_full_size_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in 0 1; do echo "== LSR_NDT_SPLIT=$v"; LSR_NDT_SPLIT=$v python tools/cfg5_probe.py 2>&1 | grep -v amdgpu | tail -6 | cut -c1-200; done
