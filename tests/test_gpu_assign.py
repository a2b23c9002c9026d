"""Device-side batched assignment (xm3d_linear_sum_assignment) against scipy.optimize.linear_sum_assignment: random costs,
heavily tied costs, the shapes of the matcher (Q = 50 queries, 1..16 targets, 10 decoder outputs x batch) and the size limits."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

pytestmark = pytest.mark.gpu


def _check(cost, nts, oq, ot):
    Q = cost.shape[1]
    for m, T in enumerate(nts):
        n = min(T, Q)
        q, t = oq[m, :n].cpu().numpy(), ot[m, :n].cpu().numpy()
        assert (oq[m, n:] == -1).all() and (ot[m, n:] == -1).all()
        assert len(set(q.tolist())) == n and len(set(t.tolist())) == n and t.min() >= 0 and t.max() < T   # a valid assignment
        assert (np.diff(q) > 0).all()                                                 # scipy's order: ascending query index
        C = cost[m, :, :T].cpu().double().numpy()
        ri, ci = linear_sum_assignment(C)
        want, got = C[ri, ci].sum(), C[q, t].sum()
        assert got <= want + 1e-9 * max(1.0, abs(want)), (m, got, want)               # optimal (ties: any optimal assignment)


@pytest.mark.parametrize("kind", ["random", "tied", "structured"])
def test_matches_scipy_optimum(dev, kind):
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(3)
    M, Q, Tm = 40, 50, 16
    nts = torch.randint(1, Tm + 1, (M,), generator=g)
    nts[0], nts[1] = Tm, 1
    if kind == "random":
        cost = torch.randn(M, Q, Tm, generator=g) * 5
    elif kind == "tied":
        cost = torch.randint(0, 3, (M, Q, Tm), generator=g).float()
    else:  # matcher-like: 5*bce + 5*dice - 2*prob, many near-equal entries
        cost = torch.rand(M, Q, 1, generator=g) * 10 + torch.rand(M, 1, Tm, generator=g) + 1e-3 * torch.randn(M, Q, Tm, generator=g)
    oq, ot = ops.linear_sum_assignment(cost.to(dev), nts.int().to(dev))
    _check(cost, nts.tolist(), oq, ot)


def test_size_limits_and_square(dev):
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(5)
    for Q, T in ((64, 64), (50, 50), (1, 1), (7, 3), (50, 58), (50, 200), (3, 256)):  # T > Q: every query matched (ScanNet200)
        cost = torch.randn(3, Q, T, generator=g)
        oq, ot = ops.linear_sum_assignment(cost.to(dev), torch.full((3,), T, dtype=torch.int32, device=dev))
        _check(cost, [T] * 3, oq, ot)
    with pytest.raises(Exception):
        ops.linear_sum_assignment(torch.zeros(1, 65, 2, device=dev), torch.ones(1, dtype=torch.int32, device=dev))
    with pytest.raises(Exception):
        ops.linear_sum_assignment(torch.zeros(1, 50, 257, device=dev), torch.ones(1, dtype=torch.int32, device=dev))


def test_matcher_uses_the_device_solver_without_host_copies(dev):
    from xmask3d_amd.criterion import HungarianMatcher

    g = torch.Generator().manual_seed(9)
    costs = [torch.randn(50, t, generator=g).to(dev) for t in (5, 12, 1, 16, 5)]
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")  # any device->host synchronisation raises
    try:
        pairs = HungarianMatcher.assign(costs)   # (first call of this size tuple uploads the counts: allowed, it is host->device)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    for C, (qi, ti) in zip(costs, pairs):
        assert qi.is_cuda and qi.dtype == torch.int64 and qi.numel() == C.shape[1]
        ri, ci = linear_sum_assignment(C.cpu().double().numpy())
        assert abs(C.cpu().double().numpy()[qi.cpu(), ti.cpu()].sum() - C.cpu().double().numpy()[ri, ci].sum()) < 1e-6
