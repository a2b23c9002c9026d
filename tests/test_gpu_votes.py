"""xm3d_scene_votes against the reference's formulation (run/infer.py:642-661,690-694): scene_pred[mask_2d, pred] += 1 per view on
CPU tensors, torch.max(dim=1) (first maximal class), counter != 0."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_votes_labels_and_seen_match_the_cpu_formulation(dev):
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(5)
    n_rows, n_cls, views = 50000, 19, 7
    rows, preds = [], []
    for _ in range(views):                                   # every view sees a random subset once, like mask_2d
        m = torch.rand(n_rows, generator=g) < 0.3
        m[40000:] = False                                    # rows nobody sees
        idx = torch.nonzero(m)[:, 0]
        rows.append(idx)
        preds.append(torch.randint(0, 4, (3, idx.numel()), generator=g))   # few classes: many ties
    rows, pred = torch.cat(rows), torch.cat(preds, 1)
    label, seen = ops.scene_votes(rows.to(dev), pred.to(dev).contiguous(), n_rows, n_cls)
    for k in range(3):
        table = torch.zeros(n_rows, n_cls, dtype=torch.int32)
        table.index_put_((rows, pred[k]), torch.ones(rows.numel(), dtype=torch.int32), accumulate=True)
        assert torch.equal(label[k].cpu(), torch.max(table, dim=1)[1])
    counter = torch.zeros(n_rows, dtype=torch.int32).index_put_((rows,), torch.ones(rows.numel(), dtype=torch.int32), accumulate=True)
    assert torch.equal(seen.cpu(), counter != 0) and not bool(seen[40000:].any())


def test_votes_edge_cases(dev):
    from xmask3d_amd import _lib, ops

    e = torch.empty(0, dtype=torch.int64, device=dev)
    label, seen = ops.scene_votes(e, torch.empty((2, 0), dtype=torch.int64, device=dev), 10, 5)
    assert label.shape == (2, 10) and int(label.abs().sum()) == 0 and not bool(seen.any())
    with pytest.raises(RuntimeError):
        ops.scene_votes(torch.zeros(3, dtype=torch.int64, device=dev), torch.zeros((1, 4), dtype=torch.int64, device=dev), 10, 5)
    # an out-of-range class (what index_put_ would assert on) raises the sticky device flag and is skipped
    label, seen = ops.scene_votes(torch.tensor([1, 2], device=dev), torch.tensor([[0, 7]], device=dev), 4, 5)
    assert _lib.lib().xm3d_check_flag() != 0
    assert _lib.lib().xm3d_check_flag() == 0 and seen.cpu().tolist() == [False, True, False, False]
