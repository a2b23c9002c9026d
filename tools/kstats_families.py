import csv,re,collections,sys,glob
f=sorted(glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True))[-1]
iters=float(sys.argv[2]) if len(sys.argv)>2 else 1
rows=list(csv.DictReader(open(f)))
fam=[("miopen/ck conv",r"miopen|igemm|^_ZN2ck|ck::|naive_conv|Conv|batched_transpose|SubTensor|gridwise|Im2|Col2"),("hipblaslt/rocblas",r"Cijk|rocblas"),("xm3d spconv",r"k_spconv|k_slab|k_bn_|k_kernel_map|k_affine"),("xm3d msda",r"k_msda"),("xm3d attn",r"k_attn"),("xm3d gemm/conv",r"k_gemm|k_conv3x3|k_split"),("xm3d norms",r"k_gn_|k_ln_|k_layer_norm|k_colsum|k_add_layer"),("xm3d other",r"xm3d"),("aten norm",r"layer_norm|RowwiseMoments|GroupNorm|group_norm|ComputeInternalGradients|GammaBeta|LayerNormBackward|layer_norm_grad|ComputeFusedParams"),("aten softmax",r"SoftMax|softmax"),("rocprim",r"rocprim"),("fill/copy rocclr",r"rocclr"),("optimizer",r"multi_tensor"),("aten other",r"at::native|at_cuda")]
tot=collections.OrderedDict((n,[0,0]) for n,_ in fam); tot["other"]=[0,0]
for r in rows:
    t=float(r["TotalDurationNs"])/1e6; c=int(r["Calls"])
    for n,p in fam:
        if re.search(p,r["Name"]): tot[n][0]+=t; tot[n][1]+=c; break
    else: tot["other"][0]+=t; tot["other"][1]+=c
T=sum(v[0] for v in tot.values())
print(f"{f}: total {T:.1f} ms = {T/iters:.1f} ms per iteration over {iters:.0f} iterations")
for n,(t,c) in tot.items(): print(f"  {n:22s} {t/iters:8.2f} ms/iter {100*t/T:5.1f}%  launches/iter {c/iters:.0f}")
