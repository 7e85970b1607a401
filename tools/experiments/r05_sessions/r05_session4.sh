#!/bin/bash
# Round 5, GPU session 4: new bench legs (shared-target batch, frontend stream, hoisted scalars), the frontend-stream test and the
# base-pointer getFitnessScore test; kernel stats of a single setInputTarget for the builder work.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05_s4; rm -rf $OUT; mkdir -p $OUT
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
timeout 900 python -m pytest tests/test_frontend_stream_gpu.py tests/test_host_cpu.py -x -q -m gpu -s 2>&1 | tail -15 > $OUT/pytest_a.txt
cat $OUT/pytest_a.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_s4/bench.json'))
for k in ('value','ms_per_step','cfg4_set_ms_one_gpu','cfg4_max_share_ms_block','cfg4_projected_speedup_8_gpus','cfg4_max_share_ms_planned'): print(k, d.get(k))
print('config passes', d['config'].get('derivative_passes_per_align'), d['config'].get('derivative_passes_last_scan'))
print('roofline scalars', {k:v for k,v in d['roofline'].items() if not isinstance(v,(dict,list,str))})
print('shared', json.dumps(d.get('ndt_shared_target_batch'))[:1500])
print('frontend', json.dumps(d.get('frontend_stream'))[:2500])
print('target', d.get('set_input_target',{}).get('median_ms'), 'next_rows', json.dumps(d.get('next_rows',{}).get('source_preprocess'))[:400])
print('cfg4 shares', json.dumps(d['cfg4_loop_batch'].get('projected_8gpu'))[:900])
PY
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_target && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_target -o t -- python $REPO/tools/target_probe.py > $OUT/target.stdout 2>&1; python $REPO/tools/stats_to_md.py /tmp/tr_target/t_kernel_stats.csv 30 > $OUT/target_stats.md 2>&1; python $REPO/tools/timeline.py /tmp/tr_target 150 60 > $OUT/timeline_target.txt 2>&1)
tail -3 $OUT/target.stdout; head -40 $OUT/target_stats.md | cut -c1-200
