#!/bin/bash
# PMC collection for the fitness search of a cfg-4 candidate set (nn1_ball_group_kernel and friends): separate rocprofv3 passes
# (--kernel-trace + --pmc only), one table per kernel printed.  Usage: bash tools/pmc_fit.sh [NC]
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
export NC=${1:-64}
OUT=$REPO/gpurun_out/pmc_fit
rm -rf $OUT; mkdir -p $OUT
timeout 600 python $REPO/tools/cfg4_stage_probe.py > $OUT/warmup.log 2>&1; echo "warm-up rc=$?"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $REPO/tools/cfg4_stage_probe.py > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run inst SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("sq", "inst", "fetch", "tcc"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    ncalls = collections.Counter()
    for f in glob.glob(out + "/%s/**/*kernel_trace.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)): ncalls[r["Kernel_Name"].split("(")[0][-40:]] += 1
    print("==", name)
    for k, v in agg.items():
        if any(t in k for t in ("nn1_", "nn_refine", "vg_", "fitness_")):
            print("  %-42s calls %5d  " % (k, ncalls[k]) + "  ".join("%s=%.4g" % (c, x / max(1, ncalls[k])) for c, x in sorted(v.items())))
PY
