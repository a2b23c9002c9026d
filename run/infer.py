#!/usr/bin/env python3
"""python run/infer.py --config configs/xmask3d_scannet_B15N4.yaml save_path DIR resume DIR/model/model_last.pth.tar
(entry point with the reference's flag convention, run/infer.sh:32-36)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmask3d_amd import config, driver

ap = argparse.ArgumentParser()
ap.add_argument("--config", required=True)
ap.add_argument("--scenes", type=int, default=2)
ap.add_argument("opts", nargs=argparse.REMAINDER)
a = ap.parse_args()
cfg = config.merge_cfg_from_list(config.load_cfg_from_cfg_file(a.config), a.opts)
driver.infer(cfg, scenes=a.scenes, resume=cfg.resume)
