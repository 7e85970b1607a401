#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_gicp; mkdir -p /root/repo/gpurun_out/prof_gicp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_gicp -o gicp -- python /root/repo/tools/r02_gicp_probe.py > /root/repo/gpurun_out/prof_gicp/stdout.log 2>&1; echo "rocprof rc=$?"
cd /root/repo
grep GICP gpurun_out/prof_gicp/stdout.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_gicp/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "rocprim" in n: n = "rocprim:" + (n.split("detail::")[2][:40] if n.count("detail::") > 1 else n[:50])
    print(f'{n[:80]:80s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us total {float(r["TotalDurationNs"])/1e6:8.3f} ms')
PY
