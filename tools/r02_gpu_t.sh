#!/bin/bash
bash /root/repo/tools/r02_gpu_o.sh 2>&1 | grep -E "GICP|gicp_" | head -12
cd /root/repo
for m in yield sleep; do LSR_WAIT_MODE=$m TAG=wait_$m timeout 200 python tools/r02_probe_quick.py 2>&1 | grep "cfg"; done
