#!/bin/bash
# What the driver runs at round end, in one GPU session: build check, smoke(), the whole GPU suite, one default bench line.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/round_check
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) | tee $O/smoke.log
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12) > $O/pytest.log; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
(timeout 900 python bench.py > $O/bench.json 2> $O/bench.err); echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/round_check/bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"],1), "us/pass", round(d["roofline"]["avg_launch_us"],2), "traffic", d["roofline"].get("traffic"), "valu", d["roofline"].get("valu_utilisation"))
c4=d["cfg4_loop_batch"]; print("cfg4 ms", round(c4["ms_per_candidate_set"],2), "within", c4["vs_cpu_oracle_fixture"]["within_1e-3m_1e-4rad"], "beyond", c4["vs_cpu_oracle_fixture"]["beyond"])
print("cfg5", round(d["cfg5_dense"]["median_ms"],3), "gicp", round(d["gicp_cfg3"]["median_ms"],3), d["gicp_cfg3"].get("batch_of_8",{}).get("speedup"), "gate", round(d["loop_gate"]["ms_per_search"],3), "target", round(d["set_input_target"]["median_ms"],3))
PY
