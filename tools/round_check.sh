#!/bin/bash
# What the driver runs at round end, in one GPU session, plus the cfg-4 stage timings: the whole GPU suite, the C-entry stage probe (64 and 8 candidates), one default bench line.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/round_check
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > $O/pytest.log; tail -8 $O/pytest.log | cut -c1-400
(timeout 600 python tools/cfg4_stage_probe.py 2>&1 | tail -2) | tee $O/stage64.log
(NC=8 timeout 600 python tools/cfg4_stage_probe.py 2>&1 | tail -2) | tee $O/stage8.log
(timeout 900 python bench.py > $O/bench.json 2> $O/bench.err); echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/round_check/bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"],1), "median ms", round(d["median_ms_per_step"],3), "us/pass", round(d["roofline"]["avg_launch_us"],2))
c4=d["cfg4_loop_batch"]; print("cfg4 ms", round(c4["ms_per_candidate_set"],2), "within", c4["vs_cpu_oracle_fixture"]["within_1e-3m_1e-4rad"], "same bits", c4.get("same_bits_as_one_by_one"))
print("chain", json.dumps(c4.get("chain_roofline"))[:600])
print("proj", json.dumps(c4.get("projected_8gpu"))[:900])
print("cfg5", round(d["cfg5_dense"]["median_ms"],3), d["cfg5_dense"]["avg_pass_us"], "gicp", round(d["gicp_cfg3"]["median_ms"],3), "gate", round(d["loop_gate"]["ms_per_search"],3), "target", round(d["set_input_target"]["median_ms"],3))
PY
