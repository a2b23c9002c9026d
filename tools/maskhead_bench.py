"""Fused prediction heads (csrc/maskhead.hip) against the op chain (einsum + xm3d_attn_mask_bias; MaskPooling's torch chain), 20 views.
python tools/maskhead_bench.py [views]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xmask3d_amd import ops, mask_head

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")


def ms(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cpu").manual_seed(1)
feat = torch.randn(B, 256, 128, 128, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
emb = (torch.randn(B, 50, 256, generator=g) / 16).to(dev, torch.bfloat16)
with torch.no_grad():
    for size in ((16, 16), (32, 32), (64, 64)):
        def chain():
            lg = torch.einsum("bqc,bchw->bqhw", emb, feat)
            return lg, ops.attn_mask_bias(lg, size, torch.bfloat16)
        t_chain = ms(chain)
        t_bias = ms(lambda: ops.mask_logits_bias(emb, feat, size, want_logits=False, bias_dtype=torch.bfloat16))
        t_full = ms(lambda: ops.mask_logits_bias(emb, feat, size, want_logits=True, bias_dtype=torch.bfloat16))
        print(f"target {size}: einsum + k_attn_mask {t_chain:7.1f} us | fused, bias only {t_bias:7.1f} us | fused, logits + bias {t_full:7.1f} us", flush=True)
    lg = torch.einsum("bqc,bchw->bqhw", emb, feat).contiguous()
    pool = mask_head.MaskPooling()
    t_own = ms(lambda: ops.mask_pool(lg, feat))
    with torch.enable_grad():
        t_chain = ms(lambda: pool(feat, lg))
    print(f"mask pooling: torch chain {t_chain:7.1f} us | k_mask_pool {t_own:7.1f} us", flush=True)
