#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_cfg4; mkdir -p /root/repo/gpurun_out/prof_cfg4
NC=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_cfg4 -o c4 -- python /root/repo/tools/r02_cfg4_probe.py > /root/repo/gpurun_out/prof_cfg4/stdout.log 2>&1; echo "rocprof rc=$?"
cd /root/repo
grep cfg4 gpurun_out/prof_cfg4/stdout.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_cfg4/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "rocprim" in n: n = "rocprim:" + (n.split("detail::")[2][:40] if n.count("detail::") > 1 else n[:50])
    print(f'{n[:80]:80s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us total {float(r["TotalDurationNs"])/1e6:8.3f} ms')
PY
