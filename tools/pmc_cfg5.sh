#!/bin/bash
# Round 5: why does tile staging (LSR_NDT_TABLE_MODE=3) lose to the dense global table on cfg 5?  Counter passes
# (separate rocprofv3 runs, --kernel-trace + --pmc only) on the 120k-point pass at ndt_resolution 2.0 and 1.0 for: the lane kernel on
# the dense table (production), the quad kernel on the dense table, the quad kernel with per-workgroup tiles staged in LDS.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_cfg5; rm -rf $OUT; mkdir -p $OUT
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
cd /tmp && export TMPDIR=/tmp
python $REPO/tools/cfg5_mode_probe.py > /dev/null 2>&1   # warm-up: workload cache
{
for RES in 2.0 1.0; do
for cfg in "lane_dense LSR_NDT_TABLE_MODE=0 LSR_NDT_QUAD=0" "quad_dense LSR_NDT_TABLE_MODE=0 LSR_NDT_QUAD=1" "quad_tile LSR_NDT_TABLE_MODE=3 LSR_NDT_QUAD=1"; do
  set -- $cfg; name=$1_res$RES; shift
  env RES=$RES "$@" timeout 300 python $REPO/tools/cfg5_mode_probe.py 2>&1 | tail -1
  for grp in "tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "mem FETCH_SIZE WRITE_SIZE" "sq SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
    set -- $grp; g=$1; shift
    env RES=$RES $(echo $cfg | cut -d' ' -f2-) timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name/$g -o $g -- python $REPO/tools/cfg5_mode_probe.py > $OUT/$name.$g.log 2>&1
  done
  python $REPO/tools/pmc_table.py $OUT/$name $name >> $OUT/table.md
done; done
} > $OUT/timing.txt 2>&1
cat $OUT/timing.txt; cat $OUT/table.md | cut -c1-400
