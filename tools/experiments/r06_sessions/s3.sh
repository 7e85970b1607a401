#!/bin/bash
# round 6 session 3: async map update stream variants + the reference-params tests
mkdir -p gpurun_out/r06_s3
O=gpurun_out/r06_s3
export HSA_ENABLE_IPC_MODE_LEGACY=0
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 1500 python -m pytest tests/test_reference_params_gpu.py -q -m gpu -s --durations=8 > $O/tests.log 2>&1; echo "tests rc=$?"
grep -v "amdgpu.ids" $O/tests.log | tail -25
for v in "torch 4 1" "torch 4 2" "torch 2 1" "own 4 1"; do set -- $v
LSR_BENCH_ASYNC_MAP_STREAMS=$3 LSR_BENCH_ASYNC_REG_STREAM=$1 GPU_MAX_HW_QUEUES=$2 timeout 900 python bench.py --no-cpu --candidates 0 > $O/bench_$1_$2_$3.json 2> $O/bench.err; echo "bench $v rc=$?"
python - $O/bench_$1_$2_$3.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().splitlines()[-1])
f=d['frontend_stream']; a=f['async_map_update']
print('inline: scan', round(f['scan_in_to_pose_out']['median_ms'],3), 'p90', round(f['scan_in_to_pose_out']['p90_ms'],3), 'upd', round(f['map_update_ms']['median'],3), 'amort', round(f['ms_per_scan_with_map_update_amortised'],3))
for k in ('hand_over_lag_0','hand_over_lag_1'):
    x=a[k]; print(k, 'scan', round(x['scan_in_to_pose_out']['median_ms'],3), 'p90', round(x['scan_in_to_pose_out']['p90_ms'],3), 'on_swap', round(x['scan_ms_median_on_hand_over_scans'],3), 'other', round(x['scan_ms_median_on_other_scans'],3), 'wait', round(x['hand_over_wait_ms_median'],3), 'worker', round(x['map_update_on_the_worker_ms_median'],3), 'amort', round(x['ms_per_scan_with_map_update_amortised'],3), 'serial', round(x['serial_replay_same_lag_ms_per_scan_amortised'],3), x['same_poses_as_the_serial_replay'])
PY
done
