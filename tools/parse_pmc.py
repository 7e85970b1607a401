import csv, glob, sys, collections
root = sys.argv[1]
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'ndt_eval' not in r['Kernel_Name']: continue
        key = (r['Grid_Size'], r.get('Workgroup_Size', ''))
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key, cs in agg.items():
        for c, v in cs.items():
            v = sorted(v)
            print(f.split('/')[-1][:24], "grid", key, c, "n", len(v), "median %.4g mean %.4g max %.4g" % (v[len(v)//2], sum(v)/len(v), v[-1]))
