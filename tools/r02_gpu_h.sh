#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ndt_gpu.py tests/test_full_size_gpu.py tests/test_edge_cases_gpu.py tests/test_loop_closure_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/r02_probe_a.py 2>&1 | grep -v amdgpu
