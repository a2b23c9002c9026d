"""CPU: config loader against the reference loader's own output, metric helpers against closed forms."""
import json
import math
import os

import numpy as np
import torch

from xmask3d_amd import config, metrics


def test_config_matches_reference_loader_dump(golden_dir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.load_cfg_from_cfg_file(os.path.join(root, "configs", "xmask3d_scannet_B15N4.yaml"))
    cfg = config.merge_cfg_from_list(cfg, ["save_path", "out/x", "batch_size", "8", "train_gpu", "[0,1]"])
    ref = json.load(open(os.path.join(golden_dir, "config_b15n4.json")))
    skip = {"data_root", "data_root_2d"}  # site-specific absolute paths in the upstream file
    assert set(cfg) == set(ref)
    for k, v in ref.items():
        if k not in skip:
            assert cfg[k] == v, k
    assert cfg.category_split.novel_category == [5, 9, 12, 16] and cfg.loss_weight.loss_binary == 16
    try:
        config.merge_cfg_from_list(cfg, ["batch_size", "'eight'"])
        assert False
    except ValueError:
        pass


def test_intersection_and_union_and_scores():
    pred = torch.tensor([0, 1, 1, 2, 2, 2, 3, 0])
    gt = torch.tensor([0, 1, 2, 2, 2, 255, 3, 1])
    i, u, t = metrics.intersection_and_union(pred, gt, 4, (255,))
    assert i.tolist() == [1, 1, 2, 1] and t.tolist() == [1, 2, 3, 1] and u.tolist() == [2, 3, 3, 1]
    s = metrics.open_vocab_scores(i, u, [0, 3], [1, 2])
    assert abs(s["mIoU_base"] - 0.75) < 1e-6 and abs(s["mIoU_novel"] - 0.5) < 1e-6 and abs(s["hIoU"] - 0.6) < 1e-6
    assert abs(metrics.cosine_learning_rate(1.0, 5, 10) - 0.5) < 1e-12 and metrics.poly_learning_rate(1.0, 0, 10) == 1.0
