"""The frozen nets' convolutions that run on the implicit-GEMM kernel (xm3d_conv_gemm_bf16): library convolution (MIOpen / CK through
F.conv2d, channels-last bf16, shipped find-db) vs ops.conv_gemm, at the bench's batch.  usage: python tools/conv_gemm_bench.py [views=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import xmask3d_amd  # noqa: F401
from xmask3d_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


CASES = [("vae down 128 @512->256", 128, 128, 512, 3, 2, (0, 0, 1, 1)), ("vae down 256 @256->128", 256, 256, 256, 3, 2, (0, 0, 1, 1)),
         ("vae down 512 @128->64", 512, 512, 128, 3, 2, (0, 0, 1, 1)), ("vae conv_out 512>8 @64", 512, 8, 64, 3, 1, (1, 1, 1, 1)),
         ("unet op 320 @64->32", 320, 320, 64, 3, 2, (1, 1, 1, 1)), ("unet op 640 @32->16", 640, 640, 32, 3, 2, (1, 1, 1, 1)),
         ("unet op 1280 @16->8", 1280, 1280, 16, 3, 2, (1, 1, 1, 1)), ("unet res 1280 @16", 1280, 1280, 16, 3, 1, (1, 1, 1, 1)),
         ("unet res 2560>1280 @16", 2560, 1280, 16, 3, 1, (1, 1, 1, 1)), ("unet res 1280 @8", 1280, 1280, 8, 3, 1, (1, 1, 1, 1)),
         ("unet res 2560>1280 @8", 2560, 1280, 8, 3, 1, (1, 1, 1, 1)), ("unet res 640>1280 @16", 640, 1280, 16, 3, 1, (1, 1, 1, 1)),
         ("unet up conv 1280 @16", 1280, 1280, 16, 3, 1, (1, 1, 1, 1)),
         ("unet proj_in 1x1 1280 @16", 1280, 1280, 16, 1, 1, (0, 0, 0, 0)), ("unet proj_in 1x1 1280 @8", 1280, 1280, 8, 1, 1, (0, 0, 0, 0)),
         ("unet skip 1x1 2560>1280 @16", 2560, 1280, 16, 1, 1, (0, 0, 0, 0)), ("unet skip 1x1 1920>640 @32", 1920, 640, 32, 1, 1, (0, 0, 0, 0)),
         ("unet skip 1x1 960>320 @64", 960, 320, 64, 1, 1, (0, 0, 0, 0)), ("pixdec input_proj 1x1 512>256 @32", 512, 256, 32, 1, 1, (0, 0, 0, 0))]
for name, cin, cout, h, k, stride, pad in CASES:
    x = torch.randn(B, cin, h, h, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device=dev)
    packed, tile, n32 = ops.conv_gemm_pack_weight(w)
    bpad = torch.zeros(n32, device=dev)
    bpad[:cout] = bias
    pt, pl, pb, pr = pad
    sym = pt == pb and pl == pr

    def libconv():
        xx = x if sym else F.pad(x, (pl, pr, pt, pb))
        return F.conv2d(xx, w, bias.to(torch.bfloat16), stride=stride, padding=(pt, pl) if sym else 0)

    def own():
        return ops.conv_gemm(x, packed, tile, n32, cout, k, stride, pad, bias=bpad)

    y1, y2 = libconv(), own()
    err = (y1.float() - y2.float()).abs().max().item() / y1.float().abs().max().item()
    tl, to = timeit(libconv), timeit(own)
    sweep = ""
    if os.environ.get("SWEEP"):
        ts = []
        for ks in (1, 2, 4, 8, 16):
            os.environ["XM3D_CONV_GEMM_KSPLIT"] = str(ks)
            try:
                ts.append(f"{ks}:{timeit(own):.0f}")
            except Exception:
                ts.append(f"{ks}:-")
        del os.environ["XM3D_CONV_GEMM_KSPLIT"]
        sweep = "  ksplit sweep us " + " ".join(ts)
    M = B * y1.shape[2] * y1.shape[3]
    fl = 2.0 * M * cin * k * k * cout
    nb = xmask3d_amd._lib.lib().xm3d_conv_gemm_ws_bytes(M, n32, cin * k * k, tile)
    print(f"{name:34s} library {tl:7.1f} us  conv_gemm {to:7.1f} us ({fl / to / 1e6:6.1f} TF{', split-K' if nb else ''})  x{tl / to:4.2f}  diff {err:.1e}{sweep}", flush=True)
