#!/bin/bash
# GPU session E of round 2: new bench.py (N=1 default line, 2 ranks on one device), full GPU test-suite, PMC passes,
# rocprofv3 kernel stats of the bench command.
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r02_bench_b.json 2> gpurun_out/r02_bench_b.err; echo "bench rc=$?"
LSR_BENCH_FORCE_DIST=1 timeout 400 python bench.py --gpus 2 --steps 10 --warmup 2 --candidates 6 --no-cpu > gpurun_out/r02_bench_2ranks.json 2> gpurun_out/r02_bench_2ranks.err; echo "bench 2 ranks rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_e.log 2>&1; echo "pytest rc=$?"
bash tools/pmc_ndt.sh r02 > gpurun_out/r02_pmc.log 2>&1; echo "pmc rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_bench; mkdir -p /root/repo/gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_bench -o bench -- python /root/repo/bench.py --steps 20 --warmup 5 --no-extras --no-cpu > /root/repo/gpurun_out/prof_bench/stdout.json 2> /root/repo/gpurun_out/prof_bench/stderr.log; echo "rocprof rc=$?"
cd /root/repo
tail -3 gpurun_out/r02_bench_b.err; head -c 6000 gpurun_out/r02_bench_b.json; echo
tail -3 gpurun_out/r02_bench_2ranks.err; head -c 2500 gpurun_out/r02_bench_2ranks.json; echo
tail -6 gpurun_out/r02_pytest_e.log
tail -8 gpurun_out/r02_pmc.log
