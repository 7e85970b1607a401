#!/bin/bash
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
for F in 24 0; do FIRST=$F python tools/share_probe.py 2>&1 | tail -1 | cut -c1-300; done
timeout 900 python -m pytest tests/test_ndt_gpu.py tests/test_full_size_gpu.py tests/test_gicp_gpu.py tests/test_concurrent_objects_gpu.py tests/test_multigpu_gpu.py tests/test_loop_closure_gpu.py -x -q -m gpu 2>&1 | tail -3
