#!/bin/bash
# PMC collection for the NDT derivative kernel (separate rocprofv3 runs, --kernel-trace + --pmc only), then
# tools/pmc_ndt_parse.py turns the CSVs into profiles/pmc_ndt_eval_latest.json (read by bench.py: roofline.traffic)
# and a markdown table.  Usage: bash tools/pmc_ndt.sh <round tag, e.g. r02>
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_ndt
rm -rf $OUT; mkdir -p $OUT
# warm-up outside the profiler: pages torch in and fills the workload cache, so that no counter pass pays for either
timeout 600 python $REPO/tools/trace_probe.py > $OUT/warmup.log 2>&1; echo "warm-up rc=$?"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $REPO/tools/trace_probe.py > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
# the derivative kernels' rows of every counter file, as rocprofv3 wrote them: what the markdown table and the JSON are made of
mkdir -p $OUT/csv
for g in fetch write sq lds tcc; do
  f=$(find $OUT/$g -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Kernel_Name|ndt_eval" "$f" > $OUT/csv/${TAG}_pmc_ndt_eval_$g.csv
done
cd $REPO && python tools/pmc_ndt_parse.py $OUT $TAG
