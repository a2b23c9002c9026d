"""1-ms-bin timeline of the timed window of a rocprofv3 kernel trace, one row per HSA queue: busy level and the dominant
kernel family per bin.  python tools/ktrace_timeline.py <dir> [ms_from] [ms_len]"""
import collections, csv, glob, sys
d = sys.argv[1]; t_from = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0; t_len = float(sys.argv[3]) if len(sys.argv) > 3 else 160.0
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rd = list(csv.DictReader(open(f)))
qk = "Queue_Id" if "Queue_Id" in rd[0] else "Queue_ID"
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r[qk])) for r in rd)
marks = [s for s, e, k, q in rows if "k_fnv_only" in k]
rows = [r for r in rows if marks[-2] < r[0] < marks[-1]]
t0 = rows[0][0] + int(t_from * 1e6)
def fam(k):
    if "spconv" in k or "k_slab" in k or "k_build" in k or "k_hash" in k or "k_kernel_map" in k or "k_bn" in k or "k_affine" in k: return "s"
    if "k_gn_" in k: return "g"
    if "xm3d" in k: return "x"
    if "conv" in k or "igemm" in k: return "c"
    if "attn" in k: return "a"
    if "Cijk" in k: return "m"
    if "elementwise" in k or "copy" in k.lower(): return "e"
    return "o"
nb = int(t_len)
per_q = collections.defaultdict(lambda: [collections.Counter() for _ in range(nb)])
for s, e, k, q in rows:
    if e < t0 or s > t0 + nb * 1e6: continue
    b0, b1 = max(0, int((s - t0) // 1e6)), min(nb - 1, int((e - t0) // 1e6))
    for b in range(b0, b1 + 1):
        lo, hi = max(s, t0 + b * 1e6), min(e, t0 + (b + 1) * 1e6)
        if hi > lo: per_q[q][b][fam(k)] += hi - lo
print(f"{f}: bins of 1 ms from +{t_from} ms; letter = dominant family (s sparse, g groupnorm, x other xm3d, c conv, a attention, m gemm, e elementwise, o other); upper case = bin > 60 % busy, '.' = idle")
for q, bins in per_q.items():
    line = ""
    for c in bins:
        tot = sum(c.values())
        if tot < 0.05e6: line += "."
        else:
            ch = c.most_common(1)[0][0]
            line += ch.upper() if tot > 0.6e6 else ch
    print(f"stream {q:>4s} |{line}|")
