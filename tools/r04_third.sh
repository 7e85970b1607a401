#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r04_third
rm -rf $O; mkdir -p $O
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
(timeout 900 python -m pytest tests/test_ndt_gpu.py -q 2>&1 | tail -40) > $O/pytest_ndt.log; tail -14 $O/pytest_ndt.log | cut -c1-300
(timeout 300 python tools/r04_chain_probe.py 2>&1 | tail -1) | tee -a $O/chain.log
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; REPS=2 timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$name -o $name -- python $REPO/tools/r04_chain_probe.py > $O/pmc_$name.log 2>&1; echo "$name rc=$?"; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE
run lds2 SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
run fetch FETCH_SIZE WRITE_SIZE
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r04_third"
agg=collections.defaultdict(list)
for f in glob.glob(O+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ndt_eval" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for c,v in sorted(agg.items()):
    v.sort(); vals=[x[1] for x in v]
    # last chain = last 44 launches; full-load launches = entries 2..12 of it
    last=vals[-44:]
    print("%-24s full-load launch (idx 2..12 mean) %.4g | whole last chain sum %.4g" % (c, sum(last[2:13])/11, sum(last)))
PY
