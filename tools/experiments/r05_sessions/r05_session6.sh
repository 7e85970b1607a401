#!/bin/bash
# Round 5, GPU session 6: lane-kernel diet (no branch around point_terms, ballot on the i1, 4 waves per SIMD enforced): parity, then timing
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05_s6; rm -rf $OUT; mkdir -p $OUT
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 1500 python -m pytest tests/test_ndt_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -4 > $OUT/pytest_a.txt; cat $OUT/pytest_a.txt
{
REPS=5 timeout 300 python tools/chain_probe.py 2>&1 | tail -1
for F in 0 24; do FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1; done
timeout 300 python tools/cfg5_probe.py 2>&1 | tail -3
} > $OUT/timing.txt 2>&1
cat $OUT/timing.txt | cut -c1-400
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -4 > $OUT/pytest_b.txt; cat $OUT/pytest_b.txt
