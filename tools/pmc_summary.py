"""Per-kernel sums of the counters of one rocprofv3 --pmc pass:  python tools/pmc_summary.py <dir> [name-substring]"""
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        did = row.get("Dispatch_Id")
        if (k, did) not in seen:
            seen.add((k, did)); n[k] += 1
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    if sub in k:
        print(f"{k[:80]}  launches {n[k]}")
        for c, v in sorted(acc[k].items()):
            print(f"    {c:32s} {v / max(n[k], 1):16.0f} per launch")
