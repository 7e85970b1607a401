#!/bin/bash
# GPU session B of round 2: quad kernel — parity tests first, then the A/B probe and the in-kernel timing.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ndt_gpu.py tests/test_full_size_gpu.py tests/test_edge_cases_gpu.py tests/test_loop_closure_gpu.py -m gpu -x -q > gpurun_out/r02_pytest_b.log 2>&1; echo "pytest rc=$?"
timeout 300 python tools/r02_probe_a.py > gpurun_out/r02_probe_b.log 2>&1; echo "probe rc=$?"
LSR_LIB_NAME=liblidarslam_reg_timing.so timeout 200 python tools/timing_probe.py > gpurun_out/r02_timing_b.log 2>&1; echo "timing rc=$?"
tail -25 gpurun_out/r02_pytest_b.log; cat gpurun_out/r02_probe_b.log; cat gpurun_out/r02_timing_b.log
