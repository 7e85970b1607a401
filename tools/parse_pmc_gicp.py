"""Parse the rocprofv3 counter CSVs of tools/pmc_gicp.sh: per-launch medians for every GICP kernel (K5 covariances, K6
correspondences, K7 Gauss-Newton step), written as <root>/<tag>_pmc_gicp.md and <root>/pmc_gicp_latest.json.  HBM bytes per
launch = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 correction of MI355X_MICROARCH.md's HBM section).  The kernel-trace rows of
the same runs give the launch durations (median over launches that did work: launches of a finished phase return at once)."""
import collections, csv, glob, json, os, re, sys

root, tag = sys.argv[1], sys.argv[2]
KERNELS = ("gicp_knn_wave_kernel", "gicp_cov_from_nbr_kernel", "gicp_corr_seeded_kernel", "gicp_corr_search_kernel", "gicp_corr_ball_kernel",
           "gicp_corr_pairs_kernel", "gicp_step_kernel")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(root + "/fetch/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def upper_median(v):   # launches of a phase that is not due return immediately: the working launches are the upper half
    v = sorted(v)
    return v[(3 * len(v)) // 4] if v else None


out = {"source": f"tools/pmc_gicp.sh {tag} -> profiles/{tag}_pmc_gicp.md",
       "correction": "2 x FETCH_SIZE (gfx950 counts the 128-byte requests of wide coalesced reads at 64 B) + WRITE_SIZE", "kernels": {}}
lines = ["# rocprofv3 PMC passes on the GICP kernels — " + tag, "",
         "Separate runs, `--kernel-trace --pmc <counters>` only (tools/pmc_gicp.sh on tools/gicp_probe.py: cfg 3, 21 registrations).",
         "Upper-quartile values per launch (launches whose phase is not due return at once and would pull a median down).", "",
         "| kernel | launches | us | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes (2F+W) | GB/s | VALU wave-instr | SQ_WAIT_ANY / SQ_WAVE_CYCLES | L2 hit |",
         "|---|---|---|---|---|---|---|---|---|---|"]
for k in KERNELS:
    c = agg.get(k)
    if not c:
        continue
    q = {n: upper_median(v) for n, v in c.items()}
    us = upper_median(dur.get(k, []))
    f_kb, w_kb = q.get("FETCH_SIZE") or 0.0, q.get("WRITE_SIZE") or 0.0
    b = int((2 * f_kb + w_kb) * 1024)
    hit = (q["TCC_HIT_sum"] / q["TCC_REQ_sum"]) if q.get("TCC_REQ_sum") else None
    wait = (q["SQ_WAIT_ANY"] / q["SQ_WAVE_CYCLES"]) if q.get("SQ_WAVE_CYCLES") else None
    out["kernels"][k] = {"launches": len(c.get("FETCH_SIZE", [])), "us": us, "fetch_size_kb": f_kb, "write_size_kb": w_kb, "bytes_per_launch": b,
                         "gb_per_s": (b / (us * 1e-6) / 1e9) if us else None, "SQ_INSTS_VALU": q.get("SQ_INSTS_VALU"), "wave_cycles_waiting": wait,
                         "l2_hit_rate": hit}
    lines.append(f"| `{k}` | {len(c.get('FETCH_SIZE', []))} | {us if us is None else round(us, 2)} | {f_kb:.1f} | {w_kb:.1f} | {b} | "
                 f"{'' if not us else round(b / (us * 1e-6) / 1e9, 1)} | {q.get('SQ_INSTS_VALU')} | {'' if wait is None else round(wait, 2)} | "
                 f"{'' if hit is None else round(hit, 3)} |")
# K6 = one launch per outer iteration since round 4 (gicp_corr_seeded_kernel); the three-launch form only under LSR_GICP_CORR_FUSED=0
k6_names = ("gicp_corr_seeded_kernel",) if "gicp_corr_seeded_kernel" in out["kernels"] else ("gicp_corr_ball_kernel", "gicp_corr_search_kernel", "gicp_corr_pairs_kernel")
k6 = [out["kernels"][k] for k in k6_names if k in out["kernels"]]
if k6:
    out["k6_per_outer_iteration"] = {"bytes": sum(x["bytes_per_launch"] for x in k6), "us": sum(x["us"] or 0 for x in k6), "kernels": list(k6_names)}
    lines += ["", f"K6 (seeded search + the points it cannot serve + pair records: {' + '.join(k6_names)}) per outer iteration: "
              f"{out['k6_per_outer_iteration']['bytes'] / 1e6:.2f} MB of HBM traffic in {out['k6_per_outer_iteration']['us']:.1f} us of kernels."]
open(os.path.join(root, f"{tag}_pmc_gicp.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(root, "pmc_gicp_latest.json"), "w"), indent=1)
print("\n".join(lines[-10:]))
