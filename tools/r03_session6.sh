#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r03_s8}
O=$REPO/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 900 python -m pytest tests/test_ndt_gpu.py tests/test_nn_gpu.py tests/test_full_size_gpu.py -m gpu -q 2>&1 | tail -30) > $O/pytest.log; echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-300
(timeout 600 python tools/r03_cfg4_probe.py 2>&1 | tail -4) | tee $O/cfg4_default.log | cut -c1-300
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_cfg4 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -o cfg4 -- python $REPO/tools/r03_cfg4_probe.py > $O/prof.stdout 2> $O/prof.stderr); echo "rocprof rc=$?"
python tools/stats_to_md.py /tmp/prof_cfg4/cfg4_kernel_stats.csv 32 > $O/cfg4_kernel_stats.md 2>&1; head -12 $O/cfg4_kernel_stats.md | cut -c1-160
rm -f $O/prof.stderr
bash tools/pmc_ndt.sh r03 > $O/pmc.log 2>&1; tail -5 $O/pmc.log | cut -c1-400
cp gpurun_out/pmc_ndt/r03_pmc_ndt_eval.md gpurun_out/pmc_ndt/pmc_ndt_eval_latest.json $O/ 2>/dev/null
cp gpurun_out/pmc_ndt/pmc_ndt_eval_latest.json profiles/pmc_ndt_eval_latest.json 2>/dev/null
(timeout 900 python bench.py > $O/bench.json 2> $O/bench.err); echo "bench rc=$?"
python - <<'PY'
import json,os
try:
    d=json.loads(open("gpurun_out/%s/bench.json" % os.environ.get("TAG","r03_s8")).read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k:d["roofline"].get(k) for k in ("bound","avg_launch_us","frac","traffic","valu_utilisation","wave_cycles_waiting","lds_bank_conflict_cycles_per_lds_active_cycle")})
    c4=d.get("cfg4_loop_batch",{}); print("cfg4", c4.get("value"), c4.get("ms_per_candidate_set"))
    c5=d.get("cfg5_dense",{}); print("cfg5", {k:c5.get(k) for k in ("median_ms","avg_pass_us","set_input_target_ms","traffic","reference_resolutions")})
except Exception as e:
    print("bench parse failed", e)
PY
