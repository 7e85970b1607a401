#!/bin/bash
# PMC passes (separate rocprofv3 runs, --kernel-trace + --pmc only) on the GICP kernels of one cfg-3 registration stream
# (tools/gicp_probe.py), then tools/parse_pmc_gicp.py -> <tag>_pmc_gicp.md + pmc_gicp_latest.json.
# Usage: bash tools/pmc_gicp.sh <round tag>
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_gicp
rm -rf $OUT; mkdir -p $OUT
timeout 600 python $REPO/tools/gicp_probe.py > $OUT/warmup.log 2>&1; echo "warm-up rc=$?"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $REPO/tools/gicp_probe.py > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
mkdir -p $OUT/csv   # the GICP kernels' rows of every counter file, as rocprofv3 wrote them
for g in fetch write sq tcc; do
  f=$(find $OUT/$g -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Kernel_Name|gicp_|nn1_|nnb_|xs_" "$f" > $OUT/csv/${TAG}_pmc_gicp_$g.csv
done
cd $REPO && python tools/parse_pmc_gicp.py $OUT $TAG
