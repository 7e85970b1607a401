#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r03_s12}
O=$REPO/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
for c in 5 5; do
  (timeout 300 python tools/r02_gicp_probe.py 2>&1 | tail -1 | cut -c1-200) | tee -a $O/gicp.log
done
(timeout 900 python -m pytest tests/test_gicp_gpu.py -m gpu -q 2>&1 | tail -5) > $O/pytest.log; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_g && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o g -- python $REPO/tools/r02_gicp_probe.py > $O/prof.stdout 2> $O/prof.stderr); echo "rocprof rc=$?"
python tools/stats_to_md.py /tmp/prof_g/g_kernel_stats.csv 12 > $O/gicp_kernel_stats.md 2>&1; cat $O/gicp_kernel_stats.md | cut -c1-160
rm -f $O/prof.stderr
