#!/bin/bash
# Round 5, GPU session 3: early neighbour-grid refinement (forked after the scatter of a small set's target builds), wider list
# launch for small fitness groups, init launch: tests of the touched paths, then the share probe A/B.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05_s3; rm -rf $OUT; mkdir -p $OUT
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 1500 python -m pytest tests/test_ndt_gpu.py tests/test_nn_gpu.py tests/test_loop_closure_gpu.py tests/test_multigpu_gpu.py tests/test_host_cpu.py -x -q -m gpu 2>&1 | tail -5 > $OUT/pytest_a.txt
cat $OUT/pytest_a.txt
{
for F in 0 24; do FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1; done
echo "[LSR_NN_REFINE_EARLY=0]"; for F in 0 24; do LSR_NN_REFINE_EARLY=0 FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1; done
echo "[64-set]"; timeout 600 python tools/cfg4_stage_probe.py 2>&1 | tail -2
echo "[64-set LSR_NN_REFINE_EARLY=1]"; LSR_NN_REFINE_EARLY=1 timeout 600 python tools/cfg4_stage_probe.py 2>&1 | tail -2
} > $OUT/stages.txt 2>&1
cat $OUT/stages.txt | cut -c1-330
trace() { name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_$name && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -o t -- python $REPO/tools/share_probe.py > $OUT/$name.stdout 2>&1
   python $REPO/tools/timeline.py /tmp/tr_$name 300 2000 > $OUT/timeline_$name.txt 2>&1)
}
trace share24 FIRST=24 MODE=share REPS=3
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -5 > $OUT/pytest_b.txt
cat $OUT/pytest_b.txt
