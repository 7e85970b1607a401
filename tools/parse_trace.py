import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start"))
ev = [(r[2]-r[1], r[1], r[2], r[3], r[4]) for r in rows if 'ndt_eval' in r[0]]
import collections
groups = collections.OrderedDict()
for e in ev: groups.setdefault((e[3], e[4]), []).append(e)
for k, v in groups.items():
    d = sorted(x[0] for x in v)
    gaps = sorted(v[i+1][1]-v[i][2] for i in range(len(v)-1) if v[i+1][1]-v[i][2] < 100000)
    print("grid", k, "launches", len(v), "dur us: min %.1f med %.1f p90 %.1f max %.1f" % (d[0]/1e3, d[len(d)//2]/1e3, d[int(len(d)*.9)]/1e3, d[-1]/1e3),
          "| gap us: med %.1f p90 %.1f" % (gaps[len(gaps)//2]/1e3, gaps[int(len(gaps)*.9)]/1e3))
