#!/bin/bash
mkdir -p gpurun_out/r06_s4
O=gpurun_out/r06_s4
export HSA_ENABLE_IPC_MODE_LEGACY=0
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 1500 python -m pytest tests/test_reference_params_gpu.py -q -m gpu -s -k gicp > $O/tests.log 2>&1; echo "tests rc=$?"
grep "gicp gate\|passed\|failed" $O/tests.log
LSR_BENCH_DUMP_SCANS=1 timeout 900 python bench.py --no-cpu --candidates 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
grep -A5 "scan dump" $O/bench.err | cut -c1-1500
