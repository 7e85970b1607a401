#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_nn_gpu.py tests/test_gicp_gpu.py tests/test_full_size_gpu.py tests/test_loop_closure_gpu.py -m gpu -x -q 2>&1 | tail -15
NC=16 timeout 120 python tools/r02_cfg4_probe.py 2>&1 | grep cfg4
timeout 120 python tools/r02_gicp_probe.py 2>&1 | grep GICP
LSR_NN_COOP=0 timeout 120 python tools/r02_gicp_probe.py 2>&1 | grep GICP
