"""Per-launch medians of every counter of the ndt_eval kernels found under a rocprofv3 --pmc output tree, as a markdown table
(one row per kernel instantiation and launch shape).  usage: python tools/pmc_table.py <dir> <label>"""
import collections, csv, glob, re, sys
root, label = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ndt_eval" not in k:
            continue
        m = re.search(r"(ndt_eval_\w+)<([^>]*)>", k)
        name = (m.group(1) + "<" + m.group(2).replace(" ", "") + ">") if m else k[:40]
        agg[(name, r["Grid_Size"], r.get("Workgroup_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "TCC_REQ_sum", "TCC_HIT_sum",
        "TCC_MISS_sum", "FETCH_SIZE", "WRITE_SIZE"]
print("| run | kernel | grid x wg | " + " | ".join(cols) + " |")
print("|---|---|---|" + "---|" * len(cols))
for key in sorted(agg):
    med = lambda c: (sorted(agg[key][c])[len(agg[key][c]) // 2] if agg[key].get(c) else None)
    print("| %s | `%s` | %s x %s | " % (label, key[0], key[1], key[2]) + " | ".join(("%.6g" % med(c)) if med(c) is not None else "-" for c in cols) + " |")
