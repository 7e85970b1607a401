#!/bin/bash
# HBM traffic of the NDT derivative kernel: two separate PMC passes (FETCH_SIZE, WRITE_SIZE), --kernel-trace only.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python /root/repo/tools/trace_probe.py > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
python /root/repo/tools/parse_pmc.py $OUT
