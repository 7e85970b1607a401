#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_nn_gpu.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | tail -5
NC=16 timeout 120 python tools/r02_cfg4_probe.py 2>&1 | grep cfg4
