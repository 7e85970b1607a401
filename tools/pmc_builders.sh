#!/bin/bash
# Round 6: HBM traffic of the two builder chains — N1 (the frontend's source preprocessing: tools/preprocess_probe.py)
# and setInputTarget on the 10-frame submap, both builders (tools/target_probe.py) — from the FETCH_SIZE / WRITE_SIZE counters, one
# rocprofv3 pass each (--kernel-trace + --pmc only), summed per kernel over ONE call of the chain.  HBM bytes = 2 x FETCH_SIZE (gfx950
# tallies the 128-byte requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md) + WRITE_SIZE.
# Output: gpurun_out/pmc_builders/{<tag>_pmc_builders.md, csv/<tag>_pmc_builders_<probe>_<counter>.csv (the raw rows)}.
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_builders; rm -rf $OUT; mkdir -p $OUT/csv
export LSR_BENCH_CACHE_DIR=/tmp/lsr_bench_cache
cd /tmp && export TMPDIR=/tmp
for probe in preprocess target; do
  python $REPO/tools/${probe}_probe.py > $OUT/$probe.txt 2>&1   # warm-up + the host-clock figure of the chain
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${probe}_$c -o p -- python $REPO/tools/${probe}_probe.py > $OUT/${probe}_$c.log 2>&1
    f=$(find $OUT/${probe}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && grep -v "rocclr\|hipExt" "$f" > $OUT/csv/${TAG}_pmc_builders_${probe}_$c.csv
  done
done
python - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, os, re, sys
out, tag = sys.argv[1], sys.argv[2]
lines = ["# HBM traffic of the builder chains (counters) — " + tag, "",
         "`2 x FETCH_SIZE + WRITE_SIZE` per kernel launch (medians), separate rocprofv3 `--kernel-trace --pmc` passes (tools/pmc_builders.sh); raw rows: `%s_pmc_csv/%s_pmc_builders_*.csv.gz` (compacted by tools/compact_pmc_csv.py)." % (tag, tag), ""]
for probe, what in (("preprocess", "N1: lsr_set_input_source_frontend on the raw 147 443-point scan (range filter + VoxelGrid(0.2) + setInputSource)"),
                    ("target", "setInputTarget on the 661 519-point 10-frame submap: counting-sort builder (vg_*) and radix builder (grid_builder = 1: leaf_key / rs_* / leaf_*)")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(out, "csv", "%s_pmc_builders_%s_%s.csv" % (tag, probe, c))
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::|lsr::|void ", "", r["Kernel_Name"]); name = re.sub(r"\(.*$", "", name)
            agg[name][c].append(float(r["Counter_Value"]))
    lines += ["## " + what, "", "```", open(os.path.join(out, probe + ".txt")).read().strip().splitlines()[-1][:300] if os.path.exists(os.path.join(out, probe + ".txt")) else "", "```", "",
              "| kernel | launches | FETCH_SIZE KB (median) | WRITE_SIZE KB (median) | HBM bytes per launch |", "|---|---|---|---|---|"]
    tot = 0.0
    for name in sorted(agg, key=lambda k: -sum(agg[k].get("FETCH_SIZE", [0]))):
        med = lambda c: (sorted(agg[name][c])[len(agg[name][c]) // 2] if agg[name].get(c) else 0.0)
        b = (2 * med("FETCH_SIZE") + med("WRITE_SIZE")) * 1024
        lines.append("| `%s` | %d | %.1f | %.1f | %.3f MB |" % (name[:70], len(agg[name].get("FETCH_SIZE", [])), med("FETCH_SIZE"), med("WRITE_SIZE"), b / 1e6))
    lines.append("")
open(os.path.join(out, tag + "_pmc_builders.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
PY
rm -rf $OUT/*_FETCH_SIZE $OUT/*_WRITE_SIZE
