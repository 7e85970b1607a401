#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_gicp; mkdir -p /root/repo/gpurun_out/prof_gicp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_gicp -o gicp -- python /root/repo/tools/r02_gicp_probe.py > /root/repo/gpurun_out/prof_gicp/stdout.log 2>&1; echo "rocprof rc=$?"
cd /root/repo
grep GICP gpurun_out/prof_gicp/stdout.log
python tools/stats_to_md.py gpurun_out/prof_gicp/gicp_kernel_stats.csv 24
