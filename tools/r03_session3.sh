#!/bin/bash
# GPU session: GPU tests (all but full-size), cfg 4 stage breakdown under A/B switches, kernel stats.  Output: gpurun_out/$TAG/
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r03_s5}
O=$REPO/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_full_size_gpu.py 2>&1 | tail -40) > $O/pytest_a.log; echo "pytest_a rc=$?"; tail -12 $O/pytest_a.log | cut -c1-300
echo "== default"; (timeout 600 python tools/r03_cfg4_probe.py 2>&1 | tail -4) | tee $O/cfg4_default.log | cut -c1-300
echo "== fitness wave form"; (LSR_FIT_GROUP_FORM=0 timeout 600 python tools/r03_cfg4_probe.py 2>&1 | tail -4) | tee $O/cfg4_wave.log | cut -c1-300
echo "== row split, 3 WGs/CU"; (LSR_LIB_NAME=liblidarslam_reg_row_split.so LSR_NDT_WGS_PER_CU=3 timeout 600 python tools/r03_cfg4_probe.py 2>&1 | tail -4) | tee $O/cfg4_rowsplit3.log | cut -c1-300
echo "== row split, 2 WGs/CU"; (LSR_LIB_NAME=liblidarslam_reg_row_split.so timeout 600 python tools/r03_cfg4_probe.py 2>&1 | tail -4) | tee $O/cfg4_rowsplit2.log | cut -c1-300
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_cfg4 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -o cfg4 -- python $REPO/tools/r03_cfg4_probe.py > $O/prof.stdout 2> $O/prof.stderr); echo "rocprof rc=$?"
python tools/stats_to_md.py /tmp/prof_cfg4/cfg4_kernel_stats.csv 32 > $O/cfg4_kernel_stats.md 2>&1; cat $O/cfg4_kernel_stats.md | cut -c1-160
rm -f $O/prof.stderr
