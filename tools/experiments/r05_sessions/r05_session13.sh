#!/bin/bash
# hygiene: the A/B switches of the round still give green parity tests; bench stats retaken next to an un-profiled bench run
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
O=gpurun_out/s13.txt; : > $O
T="tests/test_voxelgrid_gpu.py tests/test_frontend_stream_gpu.py tests/test_loop_closure_gpu.py tests/test_ndt_gpu.py"
for E in "LSR_WAIT_MODE=yield" "LSR_WAIT_MODE=sleep" "LSR_SOURCE_SYNC=1" "LSR_VG_SORT=rocprim"; do
  echo "[$E]" >> $O; env $E timeout 900 python -m pytest -x -q -m gpu $T 2>&1 | tail -1 >> $O
done
P=gpurun_out/profiles_r05; mkdir -p $P
timeout 900 python bench.py > $P/r05_bench_final.json 2> $P/bench.err; echo "bench rc=$?" >> $O
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bench && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $OLDPWD/bench.py > $OLDPWD/$P/bench.stdout 2> /dev/null); echo "rocprof rc=$?" >> $O
cp /tmp/prof_bench/bench_kernel_stats.csv $P/r05_rocprofv3_bench_kernel_stats.csv
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py (r05, MI355X; raw CSV: r05_rocprofv3_bench_kernel_stats.csv)"; echo; python tools/stats_to_md.py /tmp/prof_bench/bench_kernel_stats.csv 30; echo; echo '```'; grep -v "^W2\|^E2\|amdgpu.ids" $P/bench.stdout | tail -8 | cut -c1-2500; echo '```'; } > $P/r05_rocprofv3_bench_stats.md
head -6 $P/r05_rocprofv3_bench_stats.md | tail -2 >> $O
python -c "
import json; d=json.load(open('$P/r05_bench_final.json')); print('value', d['value'], 'avg_launch_us', d['roofline']['avg_launch_us'], 'frontend', d['frontend_stream']['scan_in_to_pose_out']['median_ms'], 'n1', d['next_rows']['source_preprocess']['ms_device_pointcloud2_payload'], 'cfg4', d['cfg4_set_ms_one_gpu'], d['cfg4_projected_speedup_8_gpus'])" >> $O
cat $O
