#!/bin/bash
# rocprofv3 kernel stats of tools/gicp_probe.py -> gpurun_out/prof_gicp
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_gicp
rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o gicp -- python /root/repo/tools/gicp_probe.py > $OUT/stdout.log 2>&1
tail -2 $OUT/stdout.log | cut -c1-200
