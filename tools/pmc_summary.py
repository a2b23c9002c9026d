"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py <dir> <kernel substring>"""
import collections, csv, glob, sys
d, sub = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{f}: {k}: dispatches {len(v)} mean {sum(v)/len(v):.1f} min {min(v):.1f} max {max(v):.1f}")
