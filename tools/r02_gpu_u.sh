#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_gicp_gpu.py tests/test_full_size_gpu.py tests/test_ndt_gpu.py -m gpu -x -q 2>&1 | tail -12
