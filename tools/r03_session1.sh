#!/bin/bash
# GPU session 1 of round 3: full GPU test suite, A/B probe of the new table modes, a short bench.  Output: gpurun_out/r03_s1/
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/${TAG:-r03_s2}
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_full_size_gpu.py 2>&1 | tail -40) > $O/pytest_a.log; echo "pytest_a rc=$?"; tail -3 $O/pytest_a.log
(timeout 900 python tools/r03_probe.py 2>&1 | tail -60) > $O/probe.log; echo "probe rc=$?"; cat $O/probe.log | cut -c1-260
(timeout 1500 python -m pytest tests/test_full_size_gpu.py -m gpu -q 2>&1 | tail -60) > $O/pytest_full.log; echo "pytest_full rc=$?"; tail -15 $O/pytest_full.log | cut -c1-400
(timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err); echo "bench rc=$?"; python - <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/%s/bench.json" % __import__("os").environ.get("TAG", "r03_s2")).read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k:d["roofline"].get(k) for k in ("avg_launch_us","frac","achieved")})
    c4=d.get("cfg4_loop_batch",{}); print("cfg4", c4.get("value"), c4.get("ms_per_candidate_set"), c4.get("vs_cpu_oracle_fixture",{}).get("beyond"), c4.get("serial_one_by_one"))
    c5=d.get("cfg5_dense",{}); print("cfg5", {k:c5.get(k) for k in ("median_ms","avg_pass_us","set_input_target_ms","derivative_passes","newton_iterations")})
    print("gicp", d.get("gicp_cfg3",{}).get("median_ms"), "target", d.get("set_input_target",{}).get("median_ms"), "parity", d.get("parity_vs_cpu"))
except Exception as e:
    print("bench parse failed", e)
PY
