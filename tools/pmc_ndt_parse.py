"""Parse the rocprofv3 counter CSVs of tools/pmc_ndt.sh: per-launch medians of the NDT derivative kernels by launch shape,
written as gpurun_out/pmc_ndt/<tag>_pmc_ndt_eval.md and profiles-ready JSON (pmc_ndt_eval_latest.json).  HBM bytes per
launch = 2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md, HBM
section) + WRITE_SIZE, both reported in KB by rocprofv3."""
import collections, csv, glob, json, os, re, sys

root, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))   # (kernel short name, grid, wg) -> counter -> values
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ndt_eval" not in k:
            continue
        short = "ndt_eval_quad_kernel" if "quad" in k else "ndt_eval_lane_kernel"
        m = re.search(r"<\s*\d+\s*,\s*(\d+)", k)   # second template argument: where the leaf records are read from (ndt.hpp: NdtTableMode)
        tab = {"0": "dense table", "1": "compact table", "2": "LDS table", "3": "tile"}.get(m.group(1) if m else "", "?")
        agg[(short, r["Grid_Size"], r.get("Workgroup_Size", ""), tab)][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = ["# rocprofv3 PMC passes on the NDT derivative kernels — " + tag, "",
         "Separate runs, `--kernel-trace --pmc <counters>` only (tools/pmc_ndt.sh on tools/trace_probe.py). Per-launch medians.", "",
         "| kernel | records | grid (threads) | workgroup | counter | launches | median | mean |", "|---|---|---|---|---|---|---|---|"]
single = None
for key in sorted(agg, key=lambda k: (k[0], int(k[1]))):
    for c, v in sorted(agg[key].items()):
        v = sorted(v)
        lines.append(f"| {key[0]} | {key[3]} | {key[1]} | {key[2]} | {c} | {len(v)} | {v[len(v) // 2]:.6g} | {sum(v) / len(v):.6g} |")
    if key[0] == "ndt_eval_quad_kernel" and "FETCH_SIZE" in agg[key] and single is None:
        single = key
if single is None:   # one-lane kernel only
    cands = [k for k in agg if "FETCH_SIZE" in agg[k]]
    single = min(cands, key=lambda k: int(k[1])) if cands else None
out = {}
if single is not None:
    med = lambda c: sorted(agg[single][c])[len(agg[single][c]) // 2] if c in agg[single] else None
    fetch_kb, write_kb = med("FETCH_SIZE"), med("WRITE_SIZE") or 0.0
    out = {"kernel": f"{single[0]} grid {single[1]} x workgroup {single[2]} (single 30k-pt registration)",
           "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
           "bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024),
           "correction": "2 x FETCH_SIZE (gfx950 counts the 128-byte requests of wide coalesced reads at 64 B) + WRITE_SIZE",
           "source": f"tools/pmc_ndt.sh {tag} -> profiles/{tag}_pmc_ndt_eval.md"}
    for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT",
              "SQ_ACTIVE_INST_LDS", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"):
        if med(c) is not None:
            out[c] = med(c)
    lines += ["", f"HBM bytes per launch of `{out['kernel']}`: 2 x {fetch_kb:.1f} KB + {write_kb:.1f} KB = {out['bytes_per_launch'] / 1e6:.3f} MB"]
    if out.get("SQ_WAVE_CYCLES"):
        lines.append(f"SQ_WAIT_ANY / SQ_WAVE_CYCLES = {100 * out['SQ_WAIT_ANY'] / out['SQ_WAVE_CYCLES']:.0f} % ; "
                     f"LDS bank-conflict cycles / LDS active cycles = {100 * out.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, out.get('SQ_ACTIVE_INST_LDS', 1.0)):.1f} %")
# the cfg-5 pass (the lane kernel on the 120 000-point scan, records in a dense global table) and the batch pass (the lane kernel on
# a set of 16 registrations, records in the LDS table)
def _entry(key, what):
    medk = lambda c: sorted(agg[key][c])[len(agg[key][c]) // 2] if c in agg[key] else None
    fkb, wkb = medk("FETCH_SIZE"), medk("WRITE_SIZE") or 0.0
    e = {"kernel": f"{key[0]} grid {key[1]} x workgroup {key[2]} ({what})", "fetch_size_kb": fkb, "write_size_kb": wkb,
         "bytes_per_launch": int((2.0 * fkb + wkb) * 1024)}
    for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT",
              "SQ_ACTIVE_INST_LDS", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"):
        if medk(c) is not None:
            e[c] = medk(c)
    return e
lanes = [k for k in agg if k[0] == "ndt_eval_lane_kernel" and "FETCH_SIZE" in agg[k]]
c5 = [k for k in lanes if k[3] != "LDS table"]
if c5:
    big = max(c5, key=lambda k: int(k[1]))
    out["cfg5"] = _entry(big, "single 120k-pt registration, dense global table")
    lines += ["", f"HBM bytes per launch of `{out['cfg5']['kernel']}`: 2 x {out['cfg5']['fetch_size_kb']:.1f} KB + {out['cfg5']['write_size_kb']:.1f} KB = {out['cfg5']['bytes_per_launch'] / 1e6:.3f} MB"]
bt = [k for k in lanes if k[3] == "LDS table"]
if bt:
    big = max(bt, key=lambda k: len(agg[k]["FETCH_SIZE"]))
    out["batch"] = _entry(big, "candidate set of 16 registrations of 30k points in one launch (counter collection serialises the streams, so the set runs as one chain here), LDS table")
    lines += ["", f"HBM bytes per launch of `{out['batch']['kernel']}`: 2 x {out['batch']['fetch_size_kb']:.1f} KB + {out['batch']['write_size_kb']:.1f} KB = {out['batch']['bytes_per_launch'] / 1e6:.3f} MB (median launch of the chain)"]
open(os.path.join(root, f"{tag}_pmc_ndt_eval.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(root, "pmc_ndt_eval_latest.json"), "w"), indent=1)
print("\n".join(lines[-6:]))
print(json.dumps(out))
