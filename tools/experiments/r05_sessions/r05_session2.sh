#!/bin/bash
# Round 5, GPU session 2: after the host-side changes (init launch instead of blits, eager fitness for small sets, low-priority side
# stream): the batch / loop-closure / multi-GPU tests, then the share probe A/B.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05_s2; rm -rf $OUT; mkdir -p $OUT
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 1500 python -m pytest tests/test_ndt_gpu.py tests/test_loop_closure_gpu.py tests/test_multigpu_gpu.py tests/test_host_cpu.py -x -q -m gpu 2>&1 | tail -5 > $OUT/pytest_a.txt
cat $OUT/pytest_a.txt
{
for F in 0 24; do FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1; done
echo "[LSR_SIDE_PRIORITY=0]"; for F in 0 24; do LSR_SIDE_PRIORITY=0 FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1; done
echo "[LSR_NDT_CHAINS=1]"; LSR_NDT_CHAINS=1 FIRST=24 timeout 300 python tools/share_probe.py 2>&1 | tail -1
echo "[64-set]"; timeout 600 python tools/cfg4_stage_probe.py 2>&1 | tail -2
echo "[64-set LSR_SIDE_PRIORITY=0]"; LSR_SIDE_PRIORITY=0 timeout 600 python tools/cfg4_stage_probe.py 2>&1 | tail -2
} > $OUT/stages.txt 2>&1
cat $OUT/stages.txt | cut -c1-330
trace() { name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_$name && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -o t -- python $REPO/tools/share_probe.py > $OUT/$name.stdout 2>&1
   python $REPO/tools/timeline.py /tmp/tr_$name 300 2000 > $OUT/timeline_$name.txt 2>&1)
}
trace share24 FIRST=24 MODE=share REPS=3
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -5 > $OUT/pytest_b.txt
cat $OUT/pytest_b.txt
