#!/bin/bash
# One GPU session: default bench line, rocprofv3 kernel stats of the bench command, the RCCL path on one GPU.
# Outputs under gpurun_out/ (summaries are copied to profiles/ by hand).
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
bash tools/profile_bench.sh > gpurun_out/profile_bench.log 2>&1; echo "profile rc=$?"
MASTER_PORT=29511 LSR_BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_dist.json 2> gpurun_out/bench_dist.err; echo "dist rc=$?"
tail -c 600 gpurun_out/bench_default.err; head -c 1500 gpurun_out/bench_default.json; echo; head -c 400 gpurun_out/bench_dist.json
