"""CPU: training-loss pieces: mask_mapper against the reference's golden vectors, matcher and
point-sampled losses against closed forms."""
import os

import numpy as np
import torch

from xmask3d_amd import criterion as C


def test_mask_mapper_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "fuser.npz"))
    fuser = C.FeatureMerger(24)
    fuser.linear.weight.data = torch.from_numpy(g["W"])
    fuser.linear.bias.data = torch.from_numpy(g["b"])

    class Cfg:
        caption_contra_2d_pre = True

    t = lambda k: [torch.from_numpy(g[f"{k}{i}"]) for i in range(2)]
    with torch.no_grad():
        fused, f2d, f3d, pre = C.mask_mapper(t("x"), t("y"), t("mask"), t("emb"), t("p3d"), fuser, torch.nn.Identity(),
                                             torch.nn.Identity(), Cfg)
    for i in range(2):
        np.testing.assert_allclose(fused[i].numpy(), g[f"fused{i}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(f2d[i].numpy(), g[f"f2d{i}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(f3d[i].numpy(), g[f"f3d{i}"], rtol=0, atol=0)
        np.testing.assert_allclose(pre[i].numpy(), g[f"f2dpre{i}"], rtol=1e-5, atol=1e-6)


def test_point_sample_and_losses_closed_forms():
    torch.manual_seed(0)
    m = torch.zeros(1, 1, 4, 4)
    m[0, 0, 1, 2] = 1.0
    centre = torch.tensor([[[2.5 / 4, 1.5 / 4]]])  # (x, y) of the centre of pixel (row 1, col 2)
    assert torch.allclose(C.point_sample(m, centre, align_corners=False), torch.ones(1, 1, 1))
    big = torch.full((3, 50), 20.0)
    tgt = torch.ones(3, 50)
    assert C.dice_loss(big, tgt, 3.0).item() < 1e-3 and C.sigmoid_ce_loss(big, tgt, 3.0).item() < 1e-6
    assert abs(C.dice_loss(-big, tgt, 3.0).item() - (1 - 1 / 51)) < 1e-3
    coords = C.get_uncertain_point_coords_with_randomness(torch.randn(2, 1, 8, 8), lambda l: -l.abs(), 40, 3.0, 0.75)
    assert coords.shape == (2, 40, 2) and float(coords.min()) >= 0 and float(coords.max()) <= 1


def test_hungarian_matcher_picks_the_matching_query():
    torch.manual_seed(1)
    tgt_masks = torch.zeros(2, 32, 32)
    tgt_masks[0, :16], tgt_masks[1, 16:] = 1, 1
    pred = torch.full((1, 5, 32, 32), -10.0)
    pred[0, 3, :16], pred[0, 1, 16:] = 10, 10  # query 3 = target 0, query 1 = target 1
    logits = torch.zeros(1, 5, 4)
    idx = C.HungarianMatcher(2.0, 5.0, 5.0, 500)({"pred_logits": logits, "pred_masks": pred},
                                                    [{"labels": torch.tensor([0, 2]), "masks": tgt_masks}])
    src, tgt = idx[0]
    assert dict(zip(tgt.tolist(), src.tolist())) == {0: 3, 1: 1}


def test_sync_moments_single_process():
    from xmask3d_amd.me_compat import sync_moments

    x = torch.randn(100, 8, dtype=torch.float64)
    mean, var, n = sync_moments(x.sum(0), (x * x).sum(0), 100)
    assert torch.allclose(mean, x.mean(0)) and torch.allclose(var, x.var(0, unbiased=False)) and float(n) == 100


def test_loss_contra_gradient_reaches_the_3d_features():
    """models/utils/criterion.py:39-182: a confident mask over mostly-novel points is selected and pulls the mean pure-3D
    feature of its points towards the (detached) mask-CLIP embedding: the gradient must reach the 3D features, not the
    CLIP embedding"""
    import types

    from xmask3d_amd.criterion import Criterion

    torch.manual_seed(0)
    Q, H, W, C, Np = 4, 24, 32, 16, 400
    stub = types.SimpleNamespace(cfg=types.SimpleNamespace(mask_shape=(H, W)), contra_criterion=torch.nn.CosineSimilarity())
    masks = torch.full((1, Q, H, W), -8.0)
    masks[0, 1, :, :16] = 8.0                         # query 1 covers the left half, confidently
    x = torch.randint(0, H, (Np,))
    y = torch.randint(0, W, (Np,))
    binary_gt = torch.zeros(Np)                       # every point novel -> "novel_num > 1.8 * base_num and > 10"
    f3d = torch.randn(Np, C, requires_grad=True)
    clip = torch.randn(1, Q, C, requires_grad=True)
    outputs = {"pred_masks": masks, "fused_pred_feature": [torch.randn(Np, C)], "mask_embed": torch.randn(1, Q, C),
               "pure3d_pred_feature": [f3d], "mask_embed_clip": clip}
    losses, picked = Criterion.loss_contra(stub, [x], [y], [binary_gt], outputs)
    assert len(picked) == 1 and picked[0][1].shape[0] == 1
    loss = losses["loss_3d_contra"]
    assert 0.0 <= float(loss) <= 2.0
    loss.backward()
    covered = y < 16
    assert float(f3d.grad[covered].abs().sum()) > 0 and float(f3d.grad[~covered].abs().sum()) == 0
    assert clip.grad is None or float(clip.grad.abs().sum()) == 0


def test_matcher_assign_on_cpu_is_scipy():
    from scipy.optimize import linear_sum_assignment
    from xmask3d_amd.criterion import HungarianMatcher

    torch.manual_seed(1)
    costs = [torch.randn(50, 7), torch.randn(50, 1)]
    for C, (i, j) in zip(costs, HungarianMatcher.assign(costs)):
        ri, ci = linear_sum_assignment(C)
        assert i.tolist() == ri.tolist() and j.tolist() == ci.tolist()
