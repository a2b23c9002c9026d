"""k_gemm (csrc/gemm.hip) against torch's F.linear (hipBLASLt) on the linear layers of the bench forward, 20 views.
python tools/gemm_bench.py [views]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from xmask3d_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")


def ms(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# name, M, K, N, act, bias, residual
CASES = [
    ("unet0 to_q 320", B * 4096, 320, 320, None, False, False),
    ("unet0 qkv fused 960", B * 4096, 320, 960, None, False, False),
    ("unet0 to_out +b+res", B * 4096, 320, 320, None, True, True),
    ("unet0 geglu 320>2x1280", B * 4096, 320, 2560, "geglu", True, False),
    ("unet0 ff out 1280>320", B * 4096, 1280, 320, None, True, True),
    ("unet1 to_q 640", B * 1024, 640, 640, None, False, False),
    ("unet1 qkv fused 1920", B * 1024, 640, 1920, None, False, False),
    ("unet1 geglu 640>2x2560", B * 1024, 640, 5120, "geglu", True, False),
    ("unet1 ff out 2560>640", B * 1024, 2560, 640, None, True, True),
    ("unet2 to_q 1280", B * 256, 1280, 1280, None, False, False),
    ("unet2 geglu 1280>2x5120", B * 256, 1280, 10240, "geglu", True, False),
    ("unet2 ff out 5120>1280", B * 256, 5120, 1280, None, True, True),
    ("ctx to_k 768>320", B * 77, 768, 320, None, False, False),
    ("clip c_fc+quickgelu", B * 257, 1024, 4096, "quick_gelu", True, False),
    ("clip c_proj +res", B * 257, 4096, 1024, None, True, True),
    ("clip in_proj", B * 257, 1024, 3072, None, True, False),
    ("vae attn q 512", B * 4096, 512, 512, None, True, False),
]
for name, M, K, N, act, hb, hr in CASES:
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    wb = w.to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev) if hb else None
    bb = bias.to(torch.bfloat16) if hb else None
    nout = N // 2 if act == "geglu" else N
    res = torch.randn(M, nout, generator=g).to(dev, torch.bfloat16) if hr else None
    packed, tile = ops.gemm_pack_weight(w, act)

    def lib_chain():
        y = F.linear(x, wb, bb)
        if act == "geglu":
            y = ops.geglu(y)
        elif act == "quick_gelu":
            y = ops.quick_gelu(y) if hasattr(ops, "quick_gelu") else y * torch.sigmoid(1.702 * y)
        if res is not None:
            y = y + res
        return y

    t_own = ms(lambda: ops.gemm(x, packed, N, tile, bias=bias, act=act, residual=res))
    t_lin = ms(lambda: F.linear(x, wb, bb))
    t_chain = ms(lib_chain)
    flop = 2.0 * M * K * N
    d = (ops.gemm(x, packed, N, tile, bias=bias, act=act, residual=res).float() - lib_chain().float()).abs().max().item()
    print(f"{name:26s} M={M:6d} K={K:5d} N={N:5d}  k_gemm {t_own*1e3:8.1f} us {flop/t_own/1e9:7.1f} TF | F.linear alone {t_lin*1e3:8.1f} us {flop/t_lin/1e9:7.1f} TF"
          f" | library chain {t_chain*1e3:8.1f} us | k_gemm/chain x{t_chain/t_own:.2f} | diff {d:.1e}", flush=True)
