"""Per-launch durations of the NDT chain kernels from a rocprofv3 kernel-trace CSV (last align of tools/chain_probe.py).
A set of six or more members runs as TWO launch chains on two streams (capi.hip: run_ndt_feeder): the launches are listed per
stream; `gap` = end of a launch -> start of the next launch on the SAME stream."""
import csv, sys, glob, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        if "ndt_eval" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]),
                         r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
# split into aligns: a gap of > 200 us after everything before it starts a new set
sets, cur, hi = [], [], 0
for r in rows:
    if cur and r[0] - hi > 200000:
        sets.append(cur); cur = []; hi = 0
    cur.append(r); hi = max(hi, r[1])
if cur: sets.append(cur)
last = sets[-1]
t0 = min(r[0] for r in last)
by_stream = collections.OrderedDict()
for r in last: by_stream.setdefault(r[4], []).append(r)
print("aligns traced", len(sets), "| last: launches", len(last), "on", len(by_stream), "stream(s) | span %.3f ms | kernel time (sum) %.3f ms"
      % ((max(r[1] for r in last) - t0) / 1e6, sum(r[1] - r[0] for r in last) / 1e6))
for s, rs in by_stream.items():
    print("stream %s: %d launches, members (grid.y) %d, first start +%.1f us, last end +%.1f us" % (s, len(rs), rs[0][3], (rs[0][0] - t0) / 1e3, (rs[-1][1] - t0) / 1e3))
    print("  idx   nb   start_us   dur_us  gap_us")
    for k, r in enumerate(rs):
        gap = (rs[k + 1][0] - r[1]) / 1e3 if k + 1 < len(rs) else 0.0
        print("  %3d %4d %10.2f %8.2f %7.2f" % (k, r[2], (r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, gap))
