#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_gicp_gpu.py tests/test_full_size_gpu.py tests/test_loop_closure_gpu.py -m gpu -x -q 2>&1 | tail -12
timeout 120 python tools/r02_gicp_probe.py 2>&1 | grep GICP
LSR_GICP_BALL=0 timeout 120 python tools/r02_gicp_probe.py 2>&1 | grep GICP
