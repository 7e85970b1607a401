#!/bin/bash
# GPU session: remaining GPU tests + cfg 4 stage breakdown + kernel stats of the cfg 4 probe.  Output: gpurun_out/$TAG/
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r03_s4}
O=$REPO/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_full_size_gpu.py 2>&1 | tail -40) > $O/pytest_a.log; echo "pytest_a rc=$?"; tail -12 $O/pytest_a.log | cut -c1-300
(timeout 600 python tools/r03_cfg4_probe.py 2>&1 | tail -8) > $O/cfg4_probe.log; cat $O/cfg4_probe.log | cut -c1-300
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_cfg4 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -o cfg4 -- python $REPO/tools/r03_cfg4_probe.py > $O/prof.stdout 2> $O/prof.stderr); echo "rocprof rc=$?"
python tools/stats_to_md.py /tmp/prof_cfg4/cfg4_kernel_stats.csv 32 > $O/cfg4_kernel_stats.md 2>&1; cat $O/cfg4_kernel_stats.md | cut -c1-160
rm -f $O/prof.stderr
