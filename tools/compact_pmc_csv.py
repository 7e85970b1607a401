"""rocprofv3 counter_collection rows -> the five columns every table of profiles/ is made of (Kernel_Name with the namespaces and the
argument list dropped, Grid_Size, Workgroup_Size, Counter_Name, Counter_Value; one row per dispatch and counter, nothing averaged),
gzip-compressed.  usage: python tools/compact_pmc_csv.py <dir with *.csv> <output dir>.  Read back with
`python -c "import csv,gzip,sys; print(list(csv.DictReader(gzip.open(sys.argv[1],'rt')))[:3])" file.csv.gz` — tools/pmc_table.py and
tools/pmc_ndt_parse.py use exactly these columns."""
import csv, glob, gzip, os, re, sys
src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
for f in sorted(glob.glob(os.path.join(src, "*.csv"))):
    out = os.path.join(dst, os.path.basename(f) + ".gz")
    with open(f) as fi, gzip.open(out, "wt", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"])
        for r in csv.DictReader(fi):
            k = re.sub(r"\(anonymous namespace\)::|lsr::|^void ", "", r["Kernel_Name"])
            k = re.sub(r"\((?:[^()]|\([^()]*\))*\)\s*$", "", k).replace(" ", "")
            w.writerow([r["Dispatch_Id"], k, r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"], r["Counter_Value"]])
    print(out, os.path.getsize(out))
