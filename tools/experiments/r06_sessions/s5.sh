#!/bin/bash
mkdir -p gpurun_out/r06_s5
O=gpurun_out/r06_s5
export HSA_ENABLE_IPC_MODE_LEGACY=0
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 1500 python -m pytest tests/test_ndt_gpu.py tests/test_nn_gpu.py tests/test_gicp_gpu.py tests/test_voxelgrid_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"
tail -5 $O/tests.log
for v in hand rocprim; do
if [ $v = rocprim ]; then export LSR_TARGET_SORT=rocprim LSR_NN_SORT=rocprim; fi
timeout 900 python bench.py --no-cpu --candidates 0 --steps 10 > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?"
python - $O/bench_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().splitlines()[-1])
c=d['cfg5_dense']; print('cfg5 set_input_target_ms', round(c['set_input_target_ms'],4), {k: round(v['set_input_target_ms'],4) for k,v in c.get('reference_resolutions',{}).items() if isinstance(v,dict) and 'set_input_target_ms' in v})
g=d['gicp_cfg3']; print('gicp median', round(g['median_ms'],4), 'first incl target', round(g['first_registration_ms_incl_target_setup'],4))
print('frontend ref yaml map_update', d['frontend_stream_lidarslam_yaml']['map_update_ms'], 'loop gate ref', d['loop_gate_reference_params']['lidarslam_yaml_ndt']['ms_per_search'], d['loop_gate_reference_params']['graphbasedslam_yaml_gicp']['ms_per_search'])
PY
done
