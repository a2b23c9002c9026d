"""Which part of a linear layer's backward goes wrong when replayed from a HIP graph on this stack?  Debug aid for the training graphs.
usage: python tools/graph_reduce_probe.py [rocblas]"""
import sys, torch
if "rocblas" in sys.argv[1:]:
    torch.backends.cuda.preferred_blas_library("cublas")
dev = torch.device("cuda:0")
torch.manual_seed(0)
F = torch.nn.functional
for rows, cols, k in ((5376, 1024, 256), (5376, 96, 256), (5376, 256, 1024), (1000, 2048, 256)):
    x = torch.randn(rows, cols, device=dev)
    w = torch.randn(cols, k, device=dev, requires_grad=True)
    b = torch.zeros(cols, device=dev, requires_grad=True)
    inp = torch.randn(rows, k, device=dev, requires_grad=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            torch.autograd.grad(F.linear(inp, w, b), (inp, w, b), x)
            torch.autograd.grad(inp @ w.t() + b, (inp, w, b), x)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        pre = x.sum(0)
        gi, gw, gb = torch.autograd.grad(F.linear(inp, w, b), (inp, w, b), x)
        gi2, gw2, gb2 = torch.autograd.grad(inp @ w.t() + b, (inp, w, b), x)
    worst = dict(pre=0, gi=0, gw=0, gb=0, gi2=0, gw2=0, gb2=0)
    for it in range(8):
        x.copy_(torch.randn(rows, cols, device=dev) * (1 + it))
        g.replay()
        torch.cuda.synchronize()
        wb = x.double().sum(0).float()
        wi = (x.double() @ w.detach().double()).float()
        ww = (x.double().t() @ inp.detach().double()).float()
        for name, got, want in (("pre", pre, wb), ("gi", gi, wi), ("gw", gw, ww), ("gb", gb, wb), ("gi2", gi2, wi), ("gw2", gw2, ww), ("gb2", gb2, wb)):
            worst[name] = max(worst[name], (got - want).abs().max().item() / want.abs().max().item())
    print(f"rows {rows} cols {cols} k {k}: worst relative error over 8 replays: " + " ".join(f"{n}={v:.1e}" for n, v in worst.items()))
