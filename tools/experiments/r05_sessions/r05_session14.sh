#!/bin/bash
# widest ball the sixteen-lane fitness search reads itself (LSR_FIT_BALL_CELLS; wider ones go to the one-wave-per-query list)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
O=gpurun_out/s14.txt; : > $O
for BC in 5 2 3 4 6 8; do
  echo "[LSR_FIT_BALL_CELLS=$BC]" >> $O
  for F in 0 24; do LSR_FIT_BALL_CELLS=$BC FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1 | cut -c1-200 >> $O; done
  LSR_FIT_BALL_CELLS=$BC timeout 300 python tools/cfg4_stage_probe.py 2>&1 | tail -2 >> $O
done
cat $O
