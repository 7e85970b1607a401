import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start"))
groups = collections.OrderedDict()
for r in rows:
    nm = 'eval' if 'ndt_eval' in r[0] else ('ctrl' if 'ndt_controller' in r[0] else None)
    if nm: groups.setdefault((nm, r[3], r[4]), []).append((r[2]-r[1], r[1], r[2]))
for k, v in groups.items():
    d = sorted(x[0] for x in v)
    print(k, "launches", len(v), "dur us: min %.1f med %.1f p90 %.1f max %.1f" % (d[0]/1e3, d[len(d)//2]/1e3, d[int(len(d)*.9)]/1e3, d[-1]/1e3))
# timeline of a stretch of the batch
ev = [(r[1], r[2], 'E' if 'ndt_eval' in r[0] else 'C') for r in rows if ('ndt_eval' in r[0] or 'ndt_controller' in r[0]) and r[4] != 1 or 'ndt_controller' in r[0]]
ev = ev[200:212]
t0 = ev[0][0]
print(" ".join("%s[%.1f-%.1f]" % (e[2], (e[0]-t0)/1e3, (e[1]-t0)/1e3) for e in ev))
