#!/bin/bash
# round 6 session 2: async map update + reference parameter sets
mkdir -p gpurun_out/r06_s2
O=gpurun_out/r06_s2
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_frontend_stream_gpu.py tests/test_reference_params_gpu.py -x -q -m gpu --durations=8 > $O/tests.log 2>&1; echo "tests rc=$?"
tail -25 $O/tests.log
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_s2/bench.json').read().splitlines()[-1])
print(json.dumps(d.get('frontend_stream'),indent=1)[:6000])
print(json.dumps(d.get('frontend_stream_lidarslam_yaml'),indent=1)[:6000])
print(json.dumps(d.get('loop_gate_reference_params'),indent=1))
print('gen', d.get('workload_generation_s'))
PY
