#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
mkdir -p gpurun_out/r04_fit
(timeout 600 python -m pytest tests/test_nn_gpu.py tests/test_loop_closure_gpu.py tests/test_ndt_gpu.py -q -x 2>&1 | tail -4)
for f in 2 0; do (LSR_FIT_GROUP_FORM=$f timeout 300 python tools/r03_cfg4_stage_c.py 2>&1 | tail -2 | sed "s/^/form $f: /") | tee -a gpurun_out/r04_fit/stage.log; done
bash tools/r04_kstats.sh fit tools/r03_cfg4_stage_c.py LSR_FIT_GROUP_FORM=2 | tee gpurun_out/r04_fit/kstats.log
