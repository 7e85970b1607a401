import csv, sys
d = sys.argv[1] if len(sys.argv) > 1 else '/root/repo/gpurun_out/prof_gicp'
rows = list(csv.DictReader(open(d + '/gicp_kernel_stats.csv')))
for r in rows[:8]:
    print("%-60s calls %5s avg %9.2f us total %8.3f ms" % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
