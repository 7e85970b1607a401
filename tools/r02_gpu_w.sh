#!/bin/bash
cd /root/repo
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
LSR_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/r02_bench2_check.json 2> gpurun_out/r02_bench2_check.err; echo "bench2 rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_bench2_check.json').read().strip().splitlines()[-1])
print(d['value'], d['n_gpus'], d.get('roofline',{}).get('avg_launch_us'), json.dumps(d.get('cfg4_loop_batch'))[:600])
PY
tail -3 gpurun_out/r02_bench2_check.err
timeout 300 python -m pytest tests/test_ndt_gpu.py -m gpu -x -q -k "batch" 2>&1 | tail -3
