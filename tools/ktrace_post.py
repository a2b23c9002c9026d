"""Kernels of the fusion + post-processing section (k_mask_owner ... last k_nearest of the forward) inside bench.py's timed
window, from a rocprofv3 kernel trace.  python tools/ktrace_post.py <dir> [n]"""
import collections, csv, glob, re, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks = [s for s, e, k in rows if "k_fnv_only" in k]
rows = [r for r in rows if marks[-2] < r[0] < marks[-1]]
starts = [s for s, e, k in rows if "k_mask_owner" in k]
segs = []
for s0 in starts:
    nxt = min([s for s in starts if s > s0] + [rows[-1][1]])
    ends = [e for s, e, k in rows if s0 <= s < nxt and ("k_ns_query" in k or ("k_nearest" in k and "seg" not in k))]
    if ends:
        segs.append((s0, max(ends)))
agg = collections.defaultdict(lambda: [0, 0])
tot_span = 0
for a, b in segs:
    tot_span += b - a
    for s, e, k in rows:
        if a <= s <= b:
            k = re.sub(r"\(anonymous namespace\)::", "", k)
            k = re.sub(r"\(.*", "", k)[:100]
            agg[k][0] += 1; agg[k][1] += e - s
busy = sum(v[1] for v in agg.values())
print(f"{len(segs)} fusion+post-processing sections, mean span {tot_span/len(segs)/1e6:.2f} ms, kernel busy {busy/len(segs)/1e6:.2f} ms, "
      f"{sum(v[0] for v in agg.values())/len(segs):.0f} kernels per section (other streams' kernels inside the span included)")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
    print("  %-100s calls/sec %6.1f avg %8.1f us total/sec %8.3f ms" % (k, c / len(segs), t / c / 1e3, t / len(segs) / 1e6))
