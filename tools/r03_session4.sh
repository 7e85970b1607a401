#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r03_s6}
O=$REPO/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/pytest.log; echo "pytest rc=$?"; tail -12 $O/pytest.log | cut -c1-300
(timeout 600 python tools/r03_cfg4_probe.py 2>&1 | tail -4) | tee $O/cfg4_default.log | cut -c1-300
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_cfg4 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -o cfg4 -- python $REPO/tools/r03_cfg4_probe.py > $O/prof.stdout 2> $O/prof.stderr); echo "rocprof rc=$?"
python tools/stats_to_md.py /tmp/prof_cfg4/cfg4_kernel_stats.csv 32 > $O/cfg4_kernel_stats.md 2>&1; head -14 $O/cfg4_kernel_stats.md | cut -c1-160
rm -f $O/prof.stderr
(timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err); echo "bench rc=$?"; python - <<'PY'
import json,os
try:
    d=json.loads(open("gpurun_out/%s/bench.json" % os.environ.get("TAG","r03_s6")).read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k:d["roofline"].get(k) for k in ("avg_launch_us","frac","achieved")})
    c4=d.get("cfg4_loop_batch",{}); print("cfg4", c4.get("value"), c4.get("ms_per_candidate_set"), c4.get("vs_cpu_oracle_fixture",{}).get("beyond"), c4.get("serial_one_by_one"))
    c5=d.get("cfg5_dense",{}); print("cfg5", {k:c5.get(k) for k in ("median_ms","avg_pass_us","set_input_target_ms","derivative_passes","newton_iterations")})
    print("gicp", d.get("gicp_cfg3",{}).get("median_ms"), "target", d.get("set_input_target",{}).get("median_ms"), "parity", d.get("parity_vs_cpu"), "loop", d.get("loop_gate",{}).get("ms_per_search"))
    print("stream", {k:(v.get("median_ms") if isinstance(v,dict) else v) for k,v in d.get("scan_stream",{}).items()})
except Exception as e:
    print("bench parse failed", e)
PY
