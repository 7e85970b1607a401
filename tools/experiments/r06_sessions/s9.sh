#!/bin/bash
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
mkdir -p gpurun_out/r06_s9
timeout 600 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -s -k cfg2_properties 2>&1 | grep "idempotence\|passed\|failed"
timeout 900 python bench.py --no-cpu > gpurun_out/r06_s9/bench.json 2> gpurun_out/r06_s9/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_s9/bench.json').read().splitlines()[-1])
r=d['roofline']
print({k:v for k,v in r.items() if not isinstance(v,(dict,str))})
print('batch', {k:v for k,v in r.get('batch',{}).items() if not isinstance(v,str)})
print('cfg5', {k:v for k,v in r.get('cfg5',{}).items() if not isinstance(v,str)})
bad=[(k,v) for k,v in r.items() if isinstance(v,float) and ('frac' in k) and v>1]
print('fractions above 1:', bad)
for k in ('value','cfg4_set_ms_one_gpu','cfg4_max_share_ms_block','cfg4_max_share_ms_planned','cfg4_projected_speedup_8_gpus','shared_target_projected_speedup_8_gpus'): print(k, d.get(k))
PY
