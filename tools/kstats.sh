#!/bin/bash
# rocprofv3 --kernel-trace --stats of one python probe: prints the top kernels.  usage: kstats.sh <outdir-name> <script> [env assignments...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; SCRIPT=$2; shift 2
cd /tmp && export TMPDIR=/tmp LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
rm -rf /tmp/ks_$NAME
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$NAME -o t -- python $REPO/$SCRIPT > /tmp/ks_$NAME.log 2>&1
python - "$NAME" <<'PY'
import csv, glob, sys
name = sys.argv[1]
f = glob.glob("/tmp/ks_%s/**/*kernel_stats.csv" % name, recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
print("== %s: top kernels (calls, total ms, avg us)" % name)
for r in rows[:14]:
    print("  %-60s %6s %9.3f %9.2f" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
tail -1 /tmp/ks_$NAME.log
