"""Steady-state kernel stats from a rocprofv3 kernel trace: only dispatches after the last MIOpen naive/find kernel
(or after --skip-frac of the span).  python tools/ktrace_window.py <dir> [n]"""
import collections, csv, glob, re, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks = [s for s, e, k in rows if "k_fnv_only" in k]
if len(marks) >= 2:  # bench.py brackets its timed region with two k_fnv_only dispatches
    rows = [r for r in rows if marks[-2] < r[0] < marks[-1]]
else:
    cut = max([e for s, e, k in rows if "naive_conv" in k] + [rows[0][0]])
    rows = [r for r in rows if r[0] >= cut]
span = (rows[-1][1] - rows[0][0]) / 1e6
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in rows:
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"\(.*", "", k)[:90]
    agg[k][0] += 1; agg[k][1] += e - s
tot = sum(v[1] for v in agg.values()) / 1e6
print(f"{f}: window {span:.1f} ms, kernels {len(rows)}, busy {tot:.1f} ms ({100*tot/span:.0f}%)")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
    print("  %-90s calls %6d avg %8.1f us total %8.2f ms %5.1f%%" % (k, c, t / c / 1e3, t / 1e6, 100 * t / 1e6 / tot))
# family totals: which implementation the window's device time runs on
fam = [("xm3d k_conv3x3 (HIP, own)", r"k_conv3x3|k_gn_affine"), ("xm3d k_gemm (HIP, own)", r"k_gemm"), ("xm3d sparse conv (HIP, own)", r"k_spconv"),
       ("xm3d attention (HIP, own)", r"k_attn_fwd|k_attn_bwd|k_softmax_rows|k_attn_mask"), ("xm3d other kernels (HIP, own)", r"xm3d"),
       ("MIOpen / CK convolutions (library)", r"^_ZN2ck|ck::|igemm|naive_conv|miopen|Conv"), ("hipBLASLt / rocBLAS GEMMs (library)", r"Cijk|rocblas"),
       ("rocPRIM (sort / scan)", r"rocprim"), ("aten elementwise / copy / reduce", r"at::native|at_cuda")]
ft = collections.OrderedDict((n_, 0) for n_, _ in fam)
ft["other"] = 0
for k, (c, t) in agg.items():
    for n_, pat in fam:
        if re.search(pat, k):
            ft[n_] += t
            break
    else:
        ft["other"] += t
print("families:")
for n_, t in ft.items():
    print("  %-45s %8.2f ms %5.1f%%" % (n_, t / 1e6, 100 * t / 1e6 / tot))
