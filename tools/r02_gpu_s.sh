#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gicp_gpu.py tests/test_full_size_gpu.py tests/test_loop_closure_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q 2>&1 | tail -8
timeout 120 python tools/r02_gicp_probe.py 2>&1 | grep GICP
LSR_GICP_FUSED=0 timeout 120 python tools/r02_gicp_probe.py 2>&1 | grep GICP
for m in spin yield sleep; do LSR_WAIT_MODE=$m TAG=wait_$m timeout 200 python tools/r02_probe_quick.py 2>&1 | grep "cfg"; done
