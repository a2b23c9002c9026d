#!/usr/bin/env python3
"""python run/train.py --config configs/xmask3d_scannet_B15N4.yaml save_path DIR [resume CKPT] [epochs N ...]
(entry point with the reference's flag convention, run/train.sh:27-31; launch with torch.distributed.run for >1 GPU)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmask3d_amd import config, driver

ap = argparse.ArgumentParser()
ap.add_argument("--config", required=True)
ap.add_argument("--iters-per-epoch", type=int, default=4)
ap.add_argument("--views-per-gpu", type=int, default=2)
ap.add_argument("opts", nargs=argparse.REMAINDER)
a = ap.parse_args()
cfg = config.merge_cfg_from_list(config.load_cfg_from_cfg_file(a.config), a.opts)
driver.train(cfg, epochs=min(cfg.epochs, int(os.environ.get("XM3D_MAX_EPOCHS", cfg.epochs))), iters_per_epoch=a.iters_per_epoch,
             views_per_gpu=a.views_per_gpu, save_path=cfg.save_path, resume=cfg.resume)
