#!/bin/bash
cd /root/repo
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 600 python -m pytest tests/test_multigpu_gpu.py -m gpu -x -q 2>&1 | tail -5
bash tools/pmc_ndt.sh r02 > gpurun_out/r02_pmc.log 2>&1; tail -6 gpurun_out/r02_pmc.log | cut -c1-600
