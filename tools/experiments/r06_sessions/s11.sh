#!/bin/bash
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
mkdir -p gpurun_out/r06_s11
timeout 1500 python -m pytest tests/test_ndt_gpu.py tests/test_full_size_gpu.py tests/test_edge_cases_gpu.py tests/test_frontend_stream_gpu.py tests/test_loop_closure_gpu.py tests/test_concurrent_objects_gpu.py -x -q -m gpu 2>&1 | tail -3
for F in 24 0; do FIRST=$F python tools/share_probe.py 2>&1 | tail -1 | cut -c1-240; done
python tools/frontend_scan_probe.py 2>&1 | grep -v amdgpu | tail -1
timeout 900 python bench.py --no-cpu --no-extras --steps 50 > gpurun_out/r06_s11/bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06_s11/bench.json').read().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'], 'pass us', d['roofline']['avg_launch_us'], d['last_step_error_vs_truth'])"
