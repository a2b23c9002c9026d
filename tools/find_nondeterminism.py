"""Which module of the forward is not bit-reproducible?  Runs the eval forward twice on the same view batch with a forward hook on
EVERY module, hashes each module's output tensors exactly (integer sum of the raw bits) and prints the first modules - in call
order - whose hashes differ between the two runs.  With "isolate": every LEAF module is instead re-run on its own inputs right
inside the hook and compared with itself - that lists every intrinsically irreproducible module, not only the first one (the
differences of run-to-run comparisons propagate).  python tools/find_nondeterminism.py [fp32|bf16] [nhwc] [isolate]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d


def bits(t):
    t = t.detach().contiguous()
    if t.dtype in (torch.float32, torch.int32):
        return t.view(torch.int32).to(torch.int64).sum()
    if t.dtype in (torch.bfloat16, torch.float16, torch.int16):
        return t.view(torch.int16).to(torch.int64).sum()
    if t.dtype == torch.float64 or t.dtype == torch.int64:
        return t.view(torch.int64).sum()
    return t.to(torch.int64).sum()


def tensors_of(o):
    if torch.is_tensor(o):
        yield o
    elif isinstance(o, (list, tuple)):
        for x in o:
            yield from tensors_of(x)
    elif isinstance(o, dict):
        for x in o.values():
            yield from tensors_of(x)
    elif hasattr(o, "F") and torch.is_tensor(getattr(o, "F", None)):
        yield o.F


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    nhwc = "nhwc" in sys.argv[2:]
    isolate = "isolate" in sys.argv[2:]
    dev = torch.device("cuda:0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_cfg_from_cfg_file(os.path.join(root, "configs", "xmask3d_scannet_B15N4.yaml"))
    torch.manual_seed(5557)
    cpu = XMASK3d(cfg).eval()
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    model = pipeline.make_inference_model(cpu, dev, dt, channels_last=nhwc, graphs=False)
    names = {m: n for n, m in model.named_modules()}
    rec = []

    def hook(m, inp, out):
        hs = [bits(t) for t in tensors_of(out) if t.is_cuda and t.numel() > 0]
        if hs:
            rec.append((names.get(m, "?"), type(m).__name__, torch.stack(hs).sum()))

    busy = [False]
    flaky = {}

    def iso_hook(m, inp, out):
        if busy[0]:
            return
        busy[0] = True
        try:
            again = m(*inp)
        finally:
            busy[0] = False
        a = [t for t in tensors_of(out) if t.is_cuda and t.numel() > 0]
        b = [t for t in tensors_of(again) if t.is_cuda and t.numel() > 0]
        if len(a) != len(b) or any(x.shape != y.shape or not torch.equal(x, y) for x, y in zip(a, b)):
            key = (names.get(m, "?"), type(m).__name__)
            flaky[key] = flaky.get(key, 0) + 1

    for m in model.modules():
        if isolate:
            if not list(m.children()):
                m.register_forward_hook(iso_hook)
        else:
            m.register_forward_hook(hook)
    sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
    T = [np.diag([50.0, 50.0, 50.0, 1.0])] * 2
    runs = []
    with torch.no_grad():
        batch = pipeline.build_scene_batch(sd, [0, 3], pipeline.default_voxelizer(device=dev), T)
        for _ in range(3):
            rec.clear()
            model(batch)
            torch.cuda.synchronize()
            runs.append([(n, t, int(h)) for n, t, h in rec])
    if isolate:
        print(f"mode {mode} nhwc {nhwc}: leaf modules whose two calls on the same input differ (3 forwards): {len(flaky)}")
        for (n, t), c in sorted(flaky.items()):
            print(f"   {n} ({t}) x{c}")
        return
    print(f"mode {mode} nhwc {nhwc}: {len(runs[0])} module calls")
    for r in (1, 2):
        bad = [(i, a[0], a[1]) for i, (a, b) in enumerate(zip(runs[0], runs[r])) if a[2] != b[2]]
        print(f"run 0 vs run {r}: {len(bad)} module outputs differ; first 12 in call order:")
        for i, n, t in bad[:12]:
            print(f"   #{i} {n} ({t})")


if __name__ == "__main__":
    main()
