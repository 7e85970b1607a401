#!/bin/bash
# Round 5, GPU session 7: full GPU test suite, default bench run, cfg-5 counter passes, resident-workgroup A/B on the 64-set chain
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05_s7; rm -rf $OUT; mkdir -p $OUT
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $OUT/pytest_all.txt; cat $OUT/pytest_all.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_s7/bench.json'))
for k in ('value','ms_per_step','cfg4_set_ms_one_gpu','cfg4_max_share_ms_block','cfg4_projected_speedup_8_gpus'): print(k, d.get(k))
print('frontend', json.dumps(d.get('frontend_stream'))[:1800])
print('n1', json.dumps(d.get('next_rows',{}).get('source_preprocess'))[:500])
print('target', d.get('set_input_target',{}).get('median_ms'))
PY
{
for w in 0 2 1; do echo "[LSR_NDT_WGS_PER_CU=$w]"; LSR_NDT_WGS_PER_CU=$w REPS=5 timeout 300 python tools/chain_probe.py 2>&1 | tail -1; done
} > $OUT/wgs.txt 2>&1; cat $OUT/wgs.txt | cut -c1-300
bash tools/pmc_cfg5.sh > $OUT/pmc_cfg5.log 2>&1; tail -25 $OUT/pmc_cfg5.log | cut -c1-420
