#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench.py command; summaries are copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench_stdout.json 2> $OUT/bench_stderr.log
ls $OUT
