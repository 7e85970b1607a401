#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r03_s7}
O=$REPO/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/pytest.log; echo "pytest rc=$?"; tail -12 $O/pytest.log | cut -c1-300
t0=$(date +%s); (timeout 900 python bench.py > $O/bench.json 2> $O/bench.err); echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s"; tail -3 $O/bench.err | cut -c1-300
python - <<'PY'
import json,os
try:
    d=json.loads(open("gpurun_out/%s/bench.json" % os.environ.get("TAG","r03_s7")).read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "gen_s", d.get("workload_generation_s"), "roofline", {k:d["roofline"].get(k) for k in ("bound","avg_launch_us","frac","achieved","valu_utilisation")})
    c4=d.get("cfg4_loop_batch",{}); print("cfg4", c4.get("value"), c4.get("ms_per_candidate_set"), c4.get("vs_cpu_oracle_fixture",{}).get("beyond"), c4.get("serial_one_by_one"))
    c5=d.get("cfg5_dense",{}); print("cfg5", {k:c5.get(k) for k in ("median_ms","avg_pass_us","set_input_target_ms","derivative_passes","newton_iterations","reference_resolutions","error")})
    g=d.get("gicp_cfg3",{}); print("gicp", g.get("median_ms"), g.get("batch_of_8"), g.get("error"))
    print("target", d.get("set_input_target",{}).get("median_ms"), "parity", d.get("parity_vs_cpu"), "loop", d.get("loop_gate",{}).get("ms_per_search"))
    cb=d.get("cpu_baseline",{}); print("cpu", {k:cb.get(k) for k in ("value","cores","ms_per_registration","one_thread","reconciliation","error")})
except Exception as e:
    print("bench parse failed", e)
PY
(LSR_BENCH_FORCE_DIST=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu > $O/bench2.json 2> $O/bench2.err); echo "bench2 rc=$?"; tail -2 $O/bench2.err | cut -c1-300
python - <<'PY'
import json,os
try:
    d=json.loads(open("gpurun_out/%s/bench2.json" % os.environ.get("TAG","r03_s7")).read().strip().splitlines()[-1])
    print("2 ranks: value", d["value"], "cfg5 ranks", d.get("cfg5_dense",{}).get("ranks"), "gicp ranks", d.get("gicp_cfg3",{}).get("ranks"), "cfg4", d.get("cfg4_loop_batch",{}).get("value"))
except Exception as e:
    print("bench2 parse failed", e)
PY
(timeout 300 python tools/r02_gicp_probe.py 2>&1 | tail -2) | tee $O/gicp_probe.log | cut -c1-400
