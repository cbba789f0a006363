"""Summarise rocprofv3 --pmc counter_collection.csv for the pmt_* kernels: python scripts/pmc_report.py <dir>..."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if "pmt_" not in name: continue
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k)
            for c, v in sorted(cs.items()):
                print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
