#!/bin/bash
# One GPU session of profiler passes for profiles/: GICP kernel stats, HBM traffic of the GICP correspondence pass,
# LDS counters of the NDT derivative kernel.  PMC passes are separate runs with --kernel-trace only.
set -u
bash /root/repo/tools/prof_gicp.sh
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
run() { probe=$1; name=$2; shift 2; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python /root/repo/tools/$probe > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run gicp_probe.py gfetch FETCH_SIZE
run gicp_probe.py gwrite WRITE_SIZE
run trace_probe.py lds1 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
run trace_probe.py lds2 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
for k in gicp_corr_kernel gicp_cov_kernel gicp_gn_kernel; do echo "== $k"; python /root/repo/tools/parse_pmc_gicp.py $OUT $k; done
echo "== ndt_eval"; python /root/repo/tools/parse_pmc.py $OUT
