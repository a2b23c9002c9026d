"""1x1 convolutions of the dense branch on channels-last bf16 tensors: MIOpen / CK convolution vs the same product as a plain GEMM
(F.linear on the (pixels, channels) view - a channels-last tensor IS that matrix).  usage: python tools/conv1x1_bench.py [views=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import xmask3d_amd  # noqa: F401

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, cin, cout, h in (("unet proj 320@64", 320, 320, 64), ("unet proj 640@32", 640, 640, 32), ("unet proj 1280@16", 1280, 1280, 16), ("unet skip 960>320@64", 960, 320, 64),
                           ("unet skip 1920>640@32", 1920, 640, 32), ("unet skip 2560>1280@16", 2560, 1280, 16), ("vae qkv 512@64", 512, 512, 64), ("vae skip 128>256@256", 128, 256, 256),
                           ("vae skip 256>512@128", 256, 512, 128), ("proj conv1 512>128@128", 512, 128, 128), ("proj conv3 128>512@128", 128, 512, 128), ("proj conv1 2560>128@16", 2560, 128, 16),
                           ("proj short 1920>512@16", 1920, 512, 16), ("proj short 640>512@64", 640, 512, 64)):
    x = torch.randn(B, cin, h, h, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w2 = w.view(cout, cin)
    bias = torch.randn(cout, device=dev, dtype=torch.bfloat16)

    def lin():
        y = F.linear(x.permute(0, 2, 3, 1).reshape(-1, cin), w2, bias)
        return y.view(B, h, h, cout).permute(0, 3, 1, 2)

    tc, tcb, tl = timeit(lambda: F.conv2d(x, w)), timeit(lambda: F.conv2d(x, w, bias)), timeit(lin)
    y1, y2 = F.conv2d(x, w, bias), lin()
    assert y2.is_contiguous(memory_format=torch.channels_last) and y2.shape == y1.shape
    err = (y1.float() - y2.float()).abs().max().item() / y1.float().abs().max().item()
    fl = 2.0 * B * h * h * cin * cout
    print(f"{name:26s} conv {tc:7.1f} us  conv+bias {tcb:7.1f} us  linear+bias {tl:7.1f} us ({fl / tl / 1e6:6.1f} TF)  x{tcb / tl:4.2f}  diff {err:.1e}", flush=True)
