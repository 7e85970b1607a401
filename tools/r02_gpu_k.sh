#!/bin/bash
# PMC passes on the NDT derivative kernel, then the kernel-trace summary of the default bench run
cd /root/repo
bash tools/pmc_ndt.sh r02 > gpurun_out/r02_pmc.log 2>&1
tail -12 gpurun_out/r02_pmc.log
mkdir -p profiles
cp gpurun_out/pmc_ndt/pmc_ndt_eval_latest.json profiles/pmc_ndt_eval_latest.json
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_bench; mkdir -p /root/repo/gpurun_out/prof_bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_bench -o bench -- python /root/repo/bench.py > /root/repo/gpurun_out/prof_bench/stdout.log 2> /root/repo/gpurun_out/prof_bench/stderr.log; echo "rocprof bench rc=$?"
cd /root/repo
rm -f gpurun_out/prof_bench/bench_kernel_trace.csv   # tens of MB; the stats table is what is kept
tail -1 gpurun_out/prof_bench/stdout.log | cut -c1-1500
head -12 gpurun_out/prof_bench/bench_kernel_stats.csv | cut -c1-200
