#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) for the GICP kernels (tools/gicp_probe.py).
set -u
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_gicp
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python /root/repo/tools/gicp_probe.py > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT
