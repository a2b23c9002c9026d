"""Writes a tiny scene in the reference's on-disk ScanNet layout (see xmask3d_amd/scannet.py) into a directory: synthetic room,
z-buffered depth maps (so that the occlusion test has something to reject), JPEG colour frames, pose text files, a caption
file.  Frame ids 0 / 7 / 20 / 100 sort differently as strings and as integers; frame 7 looks out of the room (filtered)."""
import json
import os

import numpy as np
import torch

from xmask3d_amd import synthetic


def write_scene(root, name="scene0000_00", n_points=30000, seed=11):
    from PIL import Image

    rng = np.random.RandomState(seed)
    pts = synthetic.make_box_room(rng, n_points, (4.0, 3.5, 2.5), 4)
    n = len(pts)
    feats = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    labels = rng.randint(0, 20, size=n).astype(np.int64)
    labels[rng.rand(n) < 0.05] = -100
    labels[rng.rand(n) < 0.05] = 255
    d3, d2 = os.path.join(root, "3d"), os.path.join(root, "2d")
    os.makedirs(os.path.join(d3, "val"), exist_ok=True)
    torch.save((pts.astype(np.float32), feats, labels), os.path.join(d3, "val", name + "_vh_clean_2.pth"))
    pts = pts.astype(np.float32).astype(np.float64)  # what the reader will see
    poses = {"0": synthetic.camera_pose((2.0, 1.7, 1.4), 10.0), "20": synthetic.camera_pose((2.0, 1.7, 1.4), 130.0),
             "100": synthetic.camera_pose((2.0, 1.7, 1.4), 250.0), "7": synthetic.camera_pose((2.0, -30.0, 1.4), 270.0)}
    for sub in ("color", "depth", "pose"):
        os.makedirs(os.path.join(d2, name, sub), exist_ok=True)
    k = synthetic.scannet_intrinsics()
    caps = {}
    for fid, pose in poses.items():
        cam = np.linalg.inv(pose) @ np.concatenate([pts, np.ones((n, 1))], 1).T
        z = cam[2]
        ok = z > 0.05
        u = np.round(cam[0, ok] * k[0][0] / z[ok] + k[0][2]).astype(int)
        v = np.round(cam[1, ok] * k[1][1] / z[ok] + k[1][2]).astype(int)
        inside = (u >= 0) & (u < 320) & (v >= 0) & (v < 240)
        depth = np.full((240, 320), np.inf)
        np.minimum.at(depth, (v[inside], u[inside]), z[ok][inside])
        depth[~np.isfinite(depth)] = 0.0
        Image.fromarray(np.round(depth * 1000).astype(np.uint16)).save(os.path.join(d2, name, "depth", fid + ".png"))
        img = synthetic.noise_image(rng, out=320)[:240].astype(np.uint8)
        Image.fromarray(img).save(os.path.join(d2, name, "color", fid + ".jpg"), quality=90)
        np.savetxt(os.path.join(d2, name, "pose", fid + ".txt"), pose)
        caps[fid] = f"a room seen from frame {fid}"
    with open(os.path.join(root, "captions.json"), "w") as f:
        json.dump({name: caps}, f)
    return dict(data_root=d3, data_root_2d=d2, caption_path=os.path.join(root, "captions.json"), scene=name, points=pts, feats=feats,
                labels=labels, poses=poses)
