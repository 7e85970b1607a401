#!/bin/bash
# One GPU session that produces everything profiles/ holds for a round.  usage (on the GPU box): bash tools/round_profiles.sh r03
# Output: gpurun_out/profiles_<tag>/ (copy into profiles/ afterwards).  PMC passes are separate rocprofv3 runs with
# --kernel-trace only (tools/pmc_ndt.sh).
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
P=$REPO/gpurun_out/profiles_$TAG
rm -rf $P; mkdir -p $P
cd $REPO
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache   # synthetic clouds ray-cast once per session (bench.py, tools/_cache.py)
# 1. PMC passes on the derivative kernel and on the GICP kernels -> the traffic files bench.py reads (profiles/*.json)
bash tools/pmc_ndt.sh $TAG > $P/pmc.log 2>&1; tail -4 $P/pmc.log | head -3
cp gpurun_out/pmc_ndt/${TAG}_pmc_ndt_eval.md gpurun_out/pmc_ndt/pmc_ndt_eval_latest.json $P/
mkdir -p $P/${TAG}_pmc_csv; cp gpurun_out/pmc_ndt/csv/*.csv $P/${TAG}_pmc_csv/ 2>/dev/null
cp gpurun_out/pmc_ndt/pmc_ndt_eval_latest.json profiles/pmc_ndt_eval_latest.json
if [ "${LSR_PROFILE_GICP:-1}" = "1" ]; then   # (round 6: retaken on this round's build)
bash tools/pmc_gicp.sh $TAG > $P/pmc_gicp.log 2>&1; tail -3 $P/pmc_gicp.log | cut -c1-300
cp gpurun_out/pmc_gicp/${TAG}_pmc_gicp.md gpurun_out/pmc_gicp/pmc_gicp_latest.json $P/
cp gpurun_out/pmc_gicp/csv/*.csv $P/${TAG}_pmc_csv/ 2>/dev/null
cp gpurun_out/pmc_gicp/pmc_gicp_latest.json profiles/pmc_gicp_latest.json
fi
# 1b. the cfg-5 pass in its three forms (one lane per point, two waves per chunk, four lanes per point) and the two builder chains
bash tools/pmc_cfg5.sh > $P/pmc_cfg5.log 2>&1
{ echo "# cfg 5 (120 000-point scan, dense global table): the derivative pass in three forms — $TAG"; echo; echo "us per pass (hipEvents, tools/cfg5_mode_probe.py):"; echo '```'; grep "us per pass" gpurun_out/pmc_cfg5/timing.txt; echo '```'; echo; echo "Per-launch medians of the counters (separate rocprofv3 --kernel-trace --pmc passes; raw rows: ${TAG}_pmc_csv/cfg5_*.csv.gz, compacted by tools/compact_pmc_csv.py):"; echo; cat gpurun_out/pmc_cfg5/table.md; } > $P/${TAG}_pmc_cfg5.md
for f in gpurun_out/pmc_cfg5/csv/*.csv; do cp $f $P/${TAG}_pmc_csv/cfg5_$(basename $f); done
bash tools/pmc_builders.sh $TAG > $P/pmc_builders.log 2>&1
cp gpurun_out/pmc_builders/${TAG}_pmc_builders.md $P/; cp gpurun_out/pmc_builders/csv/*.csv $P/${TAG}_pmc_csv/ 2>/dev/null
# 2. default bench run (the driver's command), JSON line kept
timeout 900 python bench.py > $P/${TAG}_bench_final.json 2> $P/bench.err; echo "bench rc=$?"
# 3. kernel-trace summaries
stats() {  # name, command...
  name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$name && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o $name -- "$@" > $P/$name.stdout 2> $P/$name.stderr); echo "rocprof $name rc=$?"
  cp /tmp/prof_$name/${name}_kernel_stats.csv $P/${TAG}_rocprofv3_${name}_kernel_stats.csv   # the raw summary, as rocprofv3 wrote it
  { echo "# rocprofv3 --kernel-trace --stats -- $* ($TAG, MI355X; raw CSV: ${TAG}_rocprofv3_${name}_kernel_stats.csv)"; echo; python tools/stats_to_md.py /tmp/prof_$name/${name}_kernel_stats.csv 30; echo; echo '```'; grep -v "^W2\|^E2\|amdgpu.ids" $P/$name.stdout | tail -8 | cut -c1-2500; echo '```'; } > $P/${TAG}_rocprofv3_${name}_stats.md
}
stats bench python $REPO/bench.py
stats gicp python $REPO/tools/gicp_probe.py
stats cfg4 python $REPO/tools/cfg4_probe.py
stats target python $REPO/tools/target_probe.py
# 3b. the shared launch chain of the 64-candidate set: host-clock time and the per-launch trace (full-load launches, widened tail)
(timeout 300 python tools/chain_probe.py 2>&1 | tail -1) > $P/${TAG}_cfg4_chain.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_chain && REPS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_chain -o t -- python $REPO/tools/chain_probe.py > /dev/null 2>&1; python $REPO/tools/chain_parse.py /tmp/tr_chain >> $P/${TAG}_cfg4_chain.txt 2>&1)
(timeout 300 python tools/cfg4_stage_probe.py 2>&1 | tail -2; NC=8 timeout 300 python tools/cfg4_stage_probe.py 2>&1 | tail -2) > $P/${TAG}_cfg4_stages.txt
# 3c. one share of an 8-GPU node (8 candidates): stage medians for the fastest and the slowest share of the block partition, and the
# kernel timeline of a whole share (tools/share_probe.py, tools/timeline.py)
(for F in 0 24; do FIRST=$F timeout 300 python tools/share_probe.py 2>&1 | tail -1; done) > $P/${TAG}_cfg4_share.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_share && FIRST=24 MODE=share REPS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_share -o t -- python $REPO/tools/share_probe.py > /dev/null 2>&1; python $REPO/tools/timeline.py /tmp/tr_share 300 2000 | tail -75 >> $P/${TAG}_cfg4_share.txt 2>&1)
# 3d. N1: the frontend's source preprocessing (hand-written LSD sort) against the rocPRIM path it replaces
(for v in lsd rocprim; do echo "[LSR_VG_SORT=$v]"; LSR_VG_SORT=$v timeout 300 python tools/preprocess_probe.py 2>&1 | tail -1; done) > $P/${TAG}_n1_preprocess.txt
stats n1 python $REPO/tools/preprocess_probe.py
# 3e. ONE frontend scan at the reference's settings (payload in HBM -> lsr_set_input_source_pc2 -> lsr_align): host-clock medians and the
# kernel timeline of a scan (tools/frontend_scan_probe.py, tools/timeline.py)
(timeout 300 python tools/frontend_scan_probe.py 2>&1 | tail -1) > $P/${TAG}_frontend_scan.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_fs && REPS=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_fs -o t -- python $REPO/tools/frontend_scan_probe.py > /dev/null 2>&1; python $REPO/tools/timeline.py /tmp/tr_fs 800 60 >> $P/${TAG}_frontend_scan.txt 2>&1)
if [ "${LSR_PROFILE_MICRO:-0}" = "1" ]; then   # micro-benchmarks behind profiles/r04_pass_timeline.md (kernels unchanged since)
(timeout 300 tools/micro/boundary_probe) > $P/${TAG}_boundary_probe.txt 2>&1
(LSR_LIB_NAME=liblidarslam_reg_timing.so timeout 400 python tools/timing_probe.py 2>&1 | grep -v amdgpu.ids) > $P/${TAG}_timing_probe.txt
fi
# 4. two ranks sharing this one device (gloo): exercises the self-spawn + sharded C-ABI path
LSR_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 > $P/${TAG}_bench_2ranks_one_device.json 2> $P/bench2.err; echo "bench2 rc=$?"
rm -f $P/*.stderr
# the raw rocprofv3 trees are tens of MB each (gpurun merges at most 64 MiB back): what is kept of them is under $P
rm -rf gpurun_out/pmc_ndt gpurun_out/pmc_gicp gpurun_out/pmc_cfg5 gpurun_out/pmc_builders
du -sh $P gpurun_out
ls -la $P
