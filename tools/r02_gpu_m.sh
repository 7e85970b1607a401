#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_ndt_gpu.py tests/test_nn_gpu.py tests/test_full_size_gpu.py tests/test_multigpu_gpu.py -m gpu -x -q 2>&1 | tail -15
OWN_STREAMS=0 NC=16 timeout 120 python tools/r02_cfg4_probe.py 2>&1 | grep cfg4
NC=16 timeout 120 python tools/r02_cfg4_probe.py 2>&1 | grep cfg4
