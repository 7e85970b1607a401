#!/bin/bash
# Round 5, GPU session 5: the hand-written LSD sort behind N1 — parity tests, the frontend-stream test, N1 timing A/B.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05_s5; rm -rf $OUT; mkdir -p $OUT
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 1200 python -m pytest tests/test_voxelgrid_gpu.py tests/test_frontend_stream_gpu.py tests/test_loop_closure_gpu.py -x -q -m gpu -s 2>&1 | tail -15 > $OUT/pytest_a.txt
cat $OUT/pytest_a.txt
for v in lsd rocprim; do
  echo "[LSR_VG_SORT=$v]"
  LSR_VG_SORT=$v timeout 300 python tools/preprocess_probe.py 2>&1 | tail -4
done > $OUT/n1.txt 2>&1
cat $OUT/n1.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_n1 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_n1 -o t -- python $REPO/tools/preprocess_probe.py > $OUT/n1_prof.stdout 2>&1; python $REPO/tools/stats_to_md.py /tmp/tr_n1/t_kernel_stats.csv 30 > $OUT/n1_stats.md 2>&1; python $REPO/tools/timeline.py /tmp/tr_n1 100 60 > $OUT/timeline_n1.txt 2>&1)
head -30 $OUT/n1_stats.md | cut -c1-160; tail -30 $OUT/timeline_n1.txt | cut -c1-130
