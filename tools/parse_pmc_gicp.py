import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else '/root/repo/gpurun_out/pmc_gicp'
pat = sys.argv[2] if len(sys.argv) > 2 else 'gicp_cov_kernel'
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if pat not in r['Kernel_Name']: continue
        agg[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
    for key, cs in agg.items():
        for c, v in sorted(cs.items()):
            v = sorted(v)
            print("grid", key, "%-24s n %3d median %.4g" % (c, len(v), v[len(v) // 2]))
