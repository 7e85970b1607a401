"""Per-launch durations of the NDT chain kernels from a rocprofv3 kernel-trace CSV (last align of tools/r04_chain_probe.py)."""
import csv, sys, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        if "ndt_eval" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), r["Kernel_Name"][:40]))
rows.sort()
# split into aligns: a gap of > 200 us starts a new chain
chains, cur = [], []
for r in rows:
    if cur and r[0] - cur[-1][1] > 200000: chains.append(cur); cur = []
    cur.append(r)
if cur: chains.append(cur)
last = chains[-1]
print("chains", len(chains), "launches in last", len(last), "span %.3f ms" % ((last[-1][1] - last[0][0]) / 1e6), "kernel time %.3f ms" % (sum(r[1] - r[0] for r in last) / 1e6))
print("idx  nb   dur_us  gap_us")
for k, r in enumerate(last):
    gap = (last[k + 1][0] - r[1]) / 1e3 if k + 1 < len(last) else 0.0
    print("%3d %4d %8.2f %7.2f" % (k, r[2], (r[1] - r[0]) / 1e3, gap))
