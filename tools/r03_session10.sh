#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r03_s10}
O=$REPO/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 300 python tools/r03_gate_probe.py 2>&1 | tail -1) | tee $O/gate_default.log
(LSR_NN_FROM_GRID=0 timeout 300 python tools/r03_gate_probe.py 2>&1 | tail -1) | tee $O/gate_cloud_grid.log
(timeout 300 python tools/r03_gate_probe.py 2>&1 | tail -1) | tee -a $O/gate_default.log
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_gate && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gate -o gate -- python $REPO/tools/r03_gate_probe.py > $O/prof.stdout 2> $O/prof.stderr); echo "rocprof rc=$?"
python tools/stats_to_md.py /tmp/prof_gate/gate_kernel_stats.csv 40 > $O/gate_kernel_stats.md 2>&1; cat $O/gate_kernel_stats.md | cut -c1-160
rm -f $O/prof.stderr
