#!/bin/bash
# Round 6: the cfg-5 pass (120k-point scan, dense global table) in its three forms — ONE lane per point (production),
# TWO waves per chunk (LSR_NDT_SPLIT=1: each wave half of the per-point neighbour tree), FOUR lanes per point (the quad kernel) — at
# ndt_resolution 2.0 and 1.0: microseconds per pass (hipEvents, tools/cfg5_mode_probe.py) and counters (separate rocprofv3 runs,
# --kernel-trace + --pmc only).  Output: gpurun_out/pmc_cfg5/{timing.txt, table.md, csv/<form>_<group>.csv = the raw
# counter_collection files the table is made of}.  usage: bash tools/pmc_cfg5.sh
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_cfg5; rm -rf $OUT; mkdir -p $OUT/csv
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
cd /tmp && export TMPDIR=/tmp
python $REPO/tools/cfg5_mode_probe.py > /dev/null 2>&1   # warm-up: workload cache
{
for RES in 2.0 1.0; do
for cfg in "lane1 LSR_NDT_TABLE_MODE=0 LSR_NDT_QUAD=0 LSR_NDT_SPLIT=0" "lane_split2 LSR_NDT_TABLE_MODE=0 LSR_NDT_QUAD=0 LSR_NDT_SPLIT=1" "quad4 LSR_NDT_TABLE_MODE=0 LSR_NDT_QUAD=1"; do
  set -- $cfg; name=$1_res$RES; shift
  env RES=$RES "$@" timeout 300 python $REPO/tools/cfg5_mode_probe.py 2>&1 | tail -1
  for grp in "tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
    set -- $grp; g=$1; shift
    env RES=$RES $(echo $cfg | cut -d' ' -f2-) timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name/$g -o $g -- python $REPO/tools/cfg5_mode_probe.py > $OUT/$name.$g.log 2>&1
    f=$(find $OUT/$name/$g -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && grep -E "Kernel_Name|ndt_eval" "$f" > $OUT/csv/${name}_$g.csv   # the derivative kernel's rows, as rocprofv3 wrote them
  done
  python $REPO/tools/pmc_table.py $OUT/$name $name >> $OUT/table.md
done; done
} > $OUT/timing.txt 2>&1
rm -rf $OUT/*/tcc $OUT/*/fetch $OUT/*/write $OUT/*/sq $OUT/*/lds $OUT/*.log 2>/dev/null
cat $OUT/timing.txt; cat $OUT/table.md | cut -c1-400
