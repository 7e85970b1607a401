#!/bin/bash
# Round 5, GPU session 1: where does an 8-candidate share (cfg 4) spend its time?  Stage medians for two shares, the chain-count /
# workgroup-size knobs on the slowest share of r04's block partition (candidates 24..31), and kernel timelines of a whole share.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05_s1; rm -rf $OUT; mkdir -p $OUT
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
{
for F in 0 24; do FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1; done
echo "--- knobs on share 24"
for env in "LSR_NDT_CHAINS=1" "LSR_NDT_WORKGROUP=1024" "LSR_NDT_CHAINS=1 LSR_NDT_WORKGROUP=1024" "LSR_NDT_QUAD_BATCH_MAX=8" "LSR_NDT_WIDEN=0" "LSR_NDT_WGS_PER_CU=2" "LSR_NDT_WGS_PER_CU=1"; do
  echo "[$env]"; env $env FIRST=24 timeout 300 python tools/share_probe.py 2>&1 | tail -1
done
} > $OUT/stages.txt 2>&1
cat $OUT/stages.txt
trace() { # name, env..., -- cmd
  name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_$name && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -o t -- python $REPO/tools/share_probe.py > $OUT/$name.stdout 2>&1
   python $REPO/tools/timeline.py /tmp/tr_$name 300 500 > $OUT/timeline_$name.txt 2>&1)
}
trace share24 FIRST=24 MODE=share REPS=3
trace share24_1chain FIRST=24 MODE=share REPS=3 LSR_NDT_CHAINS=1
head -4 $OUT/timeline_share24.txt | cut -c1-1500
ls -la $OUT
