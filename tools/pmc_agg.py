"""Aggregate rocprofv3 counter_collection CSVs: mean counter value per dispatch, per qfa kernel."""
import csv, glob, re, sys, collections
prefix = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(prefix + "*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not re.match(r"(void )?k_", k): continue          # this library's kernels only (not torch's / Tensile's)
        name = k.split("(")[0].replace("void ", "")
        per[(name, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (name, did, cn), v in per.items():
        acc[name][cn].append(v)
for name in sorted(acc):
    print(name)
    for cn in sorted(acc[name]):
        v = acc[name][cn]
        print(f"    {cn:32s} mean/dispatch {sum(v)/len(v):.6g}   (n={len(v)})")
