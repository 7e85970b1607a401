"""Timeline of the LAST burst of kernels in a rocprofv3 kernel-trace directory (a burst ends with an idle gap > GAP_US, default 300):
every launch with its stream, start, duration and geometry — how the stages of a cfg-4 share (tools/share_probe.py MODE=share)
or of a single setInputTarget lie on the device.  usage: python tools/timeline.py <trace dir> [GAP_US] [max rows]"""
import csv, glob, re, sys, collections
path = sys.argv[1]
gap_ns = int(float(sys.argv[2]) * 1000) if len(sys.argv) > 2 else 300000
max_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 400
rows = []
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::|lsr::|void ", "", r["Kernel_Name"])
        name = re.sub(r"\(.*$", "", name)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", r.get("Queue_Id", "?")),
                     int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"]),
                     int(r["LDS_Block_Size"]), int(r["VGPR_Count"])))
rows.sort()
bursts, cur, hi = [], [], 0
for r in rows:
    if cur and r[0] - hi > gap_ns:
        bursts.append(cur); cur = []; hi = 0
    cur.append(r); hi = max(hi, r[1])
if cur: bursts.append(cur)
# the last burst that holds more than a handful of launches (teardown kernels follow the timed loop)
cands = [b for b in bursts if len(b) >= 5] or bursts
last = cands[-1]
t0 = last[0][0]
span = (max(r[1] for r in last) - t0) / 1e3
print("bursts %d | last: %d launches, span %.1f us, kernel time (sum) %.1f us" % (len(bursts), len(last), span, sum(r[1] - r[0] for r in last) / 1e3))
agg = collections.OrderedDict()
for r in last:
    a = agg.setdefault(r[2], [0, 0.0]); a[0] += 1; a[1] += (r[1] - r[0]) / 1e3
print("per kernel: " + " | ".join("%s x%d %.1f us" % (k[:48], v[0], v[1]) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])))
print("  start_us   dur_us stream  wgs  gy  wg   lds vgpr kernel")
for r in last[:max_rows]:
    print("%10.2f %8.2f %6s %4d %3d %4d %5d %4d %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[4], r[5], r[6], r[7], r[8], r[2][:70]))
