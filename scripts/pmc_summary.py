"""Summarise rocprofv3 --pmc csv output: per kernel, mean of each counter over dispatches."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    print("==", d)
    dur = collections.defaultdict(list)
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in dur.items():
        if "lssvr" in k or "probe" in k:
            print(f"  {k}: n={len(v)} mean {sum(v)/len(v)/1e3:.1f} us min {min(v)/1e3:.1f} us")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in cc:
        for r in csv.DictReader(open(f)):
            if "lssvr" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d2 in acc.items():
        print("  ", k)
        for c, v in d2.items():
            print(f"      {c:36s} {sum(v)/len(v):.4g}")
