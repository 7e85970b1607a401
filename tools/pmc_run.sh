#!/bin/bash
# PMC collection passes (separate runs, --kernel-trace only) for the NDT derivative kernel.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python /root/repo/tools/trace_probe.py > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
find $OUT -name "*.csv" | head -20
