#!/bin/bash
# round 6 session 1: the world > 1 path on one device (stub collectives), bench with two ranks, baseline bench N=1
mkdir -p gpurun_out/r06_s1
O=gpurun_out/r06_s1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_multigpu_gpu.py -x -q -m gpu > $O/multigpu.log 2>&1; echo "multigpu rc=$?" >> $O/multigpu.log
tail -15 $O/multigpu.log
LSR_BENCH_FORCE_DIST=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --candidates 16 > $O/bench_w2.json 2> $O/bench_w2.err; echo "bench_w2 rc=$?"
tail -c 1500 $O/bench_w2.json; tail -5 $O/bench_w2.err
timeout 1200 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench_n1 rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_s1/bench_n1.json'))
for k in ('value','ms_per_step','shared_target_scans_total','shared_target_set_ms_one_gpu','shared_target_max_share_ms','shared_target_projected_speedup_8_gpus','cfg4_set_ms_one_gpu','cfg4_max_share_ms_block','cfg4_projected_speedup_8_gpus'):
    print(k, d.get(k))
print(json.dumps(d.get('ndt_shared_target_batch',{}).get('sharded_8_projection'),indent=1))
PY
