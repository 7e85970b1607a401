#!/bin/bash
# GPU session A of round 2: A/B probe, in-kernel timing, GPU test-suite, default bench line.
mkdir -p gpurun_out
timeout 300 python tools/r02_probe_a.py > gpurun_out/r02_probe_a.log 2>&1; echo "probe rc=$?"
LSR_LIB_NAME=liblidarslam_reg_timing.so timeout 200 python tools/timing_probe.py > gpurun_out/r02_timing_a.log 2>&1; echo "timing rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_a.log 2>&1; echo "pytest rc=$?"
timeout 400 python bench.py > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err; echo "bench rc=$?"
cat gpurun_out/r02_probe_a.log; cat gpurun_out/r02_timing_a.log; tail -15 gpurun_out/r02_pytest_a.log; head -c 3000 gpurun_out/r02_bench_a.json; tail -5 gpurun_out/r02_bench_a.err
