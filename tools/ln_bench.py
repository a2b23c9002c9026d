import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from xmask3d_amd import ops
dev = torch.device("cuda:0")
def ev(f, reps=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for rows, C in ((20 * 4096, 320), (20 * 1024, 640), (20 * 256, 1280), (20 * 307, 1024), (20 * 77, 768)):
    x = torch.randn(rows, C, device=dev).to(torch.bfloat16); d = torch.randn_like(x)
    w, b = torch.ones(C, device=dev, dtype=torch.bfloat16), torch.zeros(C, device=dev, dtype=torch.bfloat16)
    print(f"rows {rows} C {C}: torch LN {ev(lambda: F.layer_norm(x, (C,), w, b)):.1f} us, torch add+LN {ev(lambda: F.layer_norm(x + d, (C,), w, b)):.1f} us | "
          f"hip LN {ev(lambda: ops.layer_norm(x, w, b)):.1f} us, hip add+LN(+sum) {ev(lambda: ops.layer_norm(x, w, b, delta=d, want_sum=True)):.1f} us")
