#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r04_second
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 900 python -m pytest tests/test_ndt_gpu.py -q -x 2>&1 | tail -30) > $O/pytest_ndt.log; tail -12 $O/pytest_ndt.log | cut -c1-300
for v in "1024 1" "512 1" "1024 0" "512 0"; do set -- $v
  (LSR_NDT_WORKGROUP=$1 LSR_NDT_WIDEN=$2 timeout 300 python tools/r04_chain_probe.py 2>&1 | tail -1) | sed "s/^/wg $1 widen $2: /" | tee -a $O/chain.log
done
cd /tmp && export TMPDIR=/tmp
for v in "1024 1" "1024 0"; do set -- $v
  rm -rf /tmp/tr; LSR_NDT_WORKGROUP=$1 LSR_NDT_WIDEN=$2 REPS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $REPO/tools/r04_chain_probe.py > /dev/null 2>&1
  python $REPO/tools/r04_chain_parse.py /tmp/tr > $O/trace_wg$1_widen$2.txt 2>&1; head -3 $O/trace_wg$1_widen$2.txt
done
