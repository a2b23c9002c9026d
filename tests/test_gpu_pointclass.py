"""xm3d_point_class (csrc/pointclass.hip) against the op chain of pipeline.postprocess_scene it replaces (run/infer.py:489-507,
556-612): labels are arg-maxima of f32 values whose summation order differs between the two (exact-f32 MFMA vs the library GEMM, one
division by |x| after the product instead of before), so they agree except at numerical near-ties: every disagreement must be a
point whose two best gated values (torch chain) are within 1e-4 relative of each other, and there may be at most 0.1 % of them."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(n, C, Q, B, seed, K=768):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(n, K, generator=g).to(dev)
    text = F.normalize(torch.randn(C, K, generator=g), dim=-1).to(dev)
    binary = (torch.rand(n, generator=g) > 0.4).long().to(dev)
    base = torch.zeros(C, dtype=torch.bool)
    base[: (2 * C) // 3] = True
    novel = ~base
    overlap = base.float()
    masks = torch.zeros(n, Q, dtype=torch.bool)
    owner = torch.randint(0, Q, (n,), generator=g)
    covered = torch.rand(n, generator=g) > 0.3
    masks[torch.arange(n)[covered], owner[covered]] = True
    vid = torch.randint(0, B, (n,), generator=g)
    open_p = torch.softmax(3 * torch.randn(B, Q, C, generator=g), -1)
    return x, text, binary, base.to(dev), novel.to(dev), overlap.to(dev), masks.to(dev), vid.to(dev), open_p.to(dev)


def _gate(v, binary, base, novel):
    return torch.where(binary.bool()[:, None], v.masked_fill(novel, -1e10), v.masked_fill(base, -1e10))


def _agree(label, values):
    ref = values.argmax(1)
    bad = label != ref
    top2 = values.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]).abs() / top2[:, 0].abs().clamp_min(1e-30)
    assert int(bad.sum()) <= max(1, label.numel() // 1000), int(bad.sum())
    assert bool((gap[bad] < 1e-4).all()), gap[bad]
    # a disagreeing label is the torch chain's runner-up, never anything else
    runner = values.topk(2, dim=1).indices[:, 1]
    assert bool((label[bad] == runner[bad]).all())


@pytest.mark.parametrize("n,C,Q,B", [(5000, 19, 50, 3), (33, 20, 7, 1), (70001, 19, 50, 20), (1, 32, 2, 1)])
def test_fused_prediction_chain(n, C, Q, B):
    from xmask3d_amd import ops

    x, text, binary, base, novel, overlap, masks, vid, open_p = _case(n, C, Q, B, seed=n + C)
    scale = torch.tensor(14.2857, device=x.device)
    br, nr = 0.65, 0.45
    with torch.no_grad():
        assert ops.point_class_supported(x, text)
        label = ops.point_class(x, text, binary, base, novel, ensemble=(scale.reshape(1), masks, vid, open_p, overlap, br, nr))
        probs = (scale * (F.normalize(x, dim=-1) @ text.t())).softmax(-1)
        po = open_p[vid, masks.to(torch.uint8).argmax(1)]
        b = (probs ** br * po ** (1 - br)).log() * overlap
        nn = (probs ** nr * po ** (1 - nr)).log() * (1 - overlap)
        val = torch.where(masks.any(1)[:, None], b + nn, probs)
        _agree(label, _gate(val, binary, base, novel))


@pytest.mark.parametrize("n,C", [(5000, 19), (70001, 19), (64, 32)])
def test_plain_and_gathered_labels(n, C):
    from xmask3d_amd import ops

    x, text, binary, base, novel, *_ = _case(n, C, 4, 1, seed=7 * n + C)
    g = torch.Generator(device="cpu").manual_seed(3)
    fill = torch.randint(0, n, (n,), generator=g).to(x.device)
    with torch.no_grad():
        _agree(ops.point_class(x, text, binary, base, novel), _gate(x @ text.t(), binary, base, novel))
        _agree(ops.point_class(x, text, binary, base, novel, row_index=fill), _gate((x @ text.t())[fill], binary, base, novel))
        # strided rows (a column block of a wider table)
        wide = torch.cat([x, x], 1)
        assert torch.equal(ops.point_class(wide[:, : x.shape[1]], text, binary, base, novel), ops.point_class(x, text, binary, base, novel))
