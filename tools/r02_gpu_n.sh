#!/bin/bash
cd /root/repo
NC=16 timeout 120 python tools/r02_cfg4_probe.py 2>&1 | grep cfg4
GPU_MAX_HW_QUEUES=8 NC=16 timeout 120 python tools/r02_cfg4_probe.py 2>&1 | grep cfg4
GPU_MAX_HW_QUEUES=16 NC=16 timeout 120 python tools/r02_cfg4_probe.py 2>&1 | grep cfg4
GPU_MAX_HW_QUEUES=2 NC=16 timeout 120 python tools/r02_cfg4_probe.py 2>&1 | grep cfg4
