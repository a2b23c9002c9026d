"""Print the top kernels of a rocprofv3 --stats run: python tools/kstats.py <dir> [n]"""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
for f in sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f, "total GPU kernel time %.3f ms" % (tot / 1e6))
    for r in rows[:n]:
        print("  %-70s calls %6s avg %10.1f us  %6.2f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
