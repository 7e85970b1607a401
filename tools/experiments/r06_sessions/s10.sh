#!/bin/bash
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
mkdir -p gpurun_out/r06_s10
timeout 900 python -m pytest tests/test_voxelgrid_gpu.py tests/test_frontend_stream_gpu.py tests/test_reference_params_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -15
for v in 0 1; do echo "== LSR_VOXEL_FILTER_EXACT=$v"; LSR_VOXEL_FILTER_EXACT=$v python tools/preprocess_probe.py 2>&1 | grep -v amdgpu | tail -2 | cut -c1-600; LSR_VOXEL_FILTER_EXACT=$v python tools/frontend_scan_probe.py 2>&1 | grep -v amdgpu | tail -1; done
