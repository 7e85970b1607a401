#!/bin/bash
# probe + kernel trace of setInputTarget
mkdir -p gpurun_out
timeout 200 python tools/r02_probe_a.py > gpurun_out/r02_probe_c.log 2>&1; echo "probe rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_target; mkdir -p /root/repo/gpurun_out/prof_target
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_target -o tgt -- python /root/repo/tools/target_probe.py > /root/repo/gpurun_out/prof_target/stdout.log 2>&1; echo "rocprof rc=$?"
cd /root/repo
grep -v amdgpu.ids gpurun_out/r02_probe_c.log
cat gpurun_out/prof_target/stdout.log | grep -v amdgpu.ids | tail -5
f=$(find gpurun_out/prof_target -name "*kernel_stats.csv" | head -1); echo $f; head -30 $f
