#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ndt_gpu.py tests/test_edge_cases_gpu.py tests/test_voxelgrid_gpu.py -m gpu -x -q > gpurun_out/r02_pytest_d.log 2>&1; echo "pytest rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_target; mkdir -p /root/repo/gpurun_out/prof_target
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_target -o tgt -- python /root/repo/tools/target_probe.py > /root/repo/gpurun_out/prof_target/stdout.log 2>&1; echo "rocprof rc=$?"
cd /root/repo
tail -8 gpurun_out/r02_pytest_d.log
grep "setInputTarget" gpurun_out/prof_target/stdout.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_target/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "rocprim" in n: n = "rocprim:" + n.split("detail::")[2][:40] if "detail::" in n else n[:50]
    print(f'{n[:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
