#!/bin/bash
# Round 5, GPU session 8: target builder changes (wave-scan cellscan / pack, large cells summed by the whole workgroup): parity + timing
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05_s8; rm -rf $OUT; mkdir -p $OUT
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 1500 python -m pytest tests/test_ndt_gpu.py tests/test_voxelgrid_gpu.py tests/test_loop_closure_gpu.py tests/test_edge_cases_gpu.py tests/test_nn_gpu.py -x -q -m gpu 2>&1 | tail -4 > $OUT/pytest_a.txt; cat $OUT/pytest_a.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_target && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_target -o t -- python $REPO/tools/target_probe.py > $OUT/target.stdout 2>&1; python $REPO/tools/stats_to_md.py /tmp/tr_target/t_kernel_stats.csv 24 > $OUT/target_stats.md 2>&1)
grep -v "^W2\|^E2\|amdgpu" $OUT/target.stdout | tail -4; grep "vg_\|lds_pack\|leaf" $OUT/target_stats.md | cut -c1-120
for F in 0 24; do FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1; done | cut -c1-330
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -4 > $OUT/pytest_b.txt; cat $OUT/pytest_b.txt
