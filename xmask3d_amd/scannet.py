"""Reader for the reference's on-disk ScanNet layout.

Counterpart of ``ScannetLoaderFull.__getitem__`` (/root/reference/dataset/data_loader_infer.py:88-308) up to the point where
the per-view samples are assembled - which ``pipeline.SceneOnDevice`` / ``build_scene_batch`` then do on the device:

    <data_root>/<split>/<scene>_vh_clean_2.pth        torch.save((coords (N,3) f32, colours (N,3) in [-1,1], labels (N,)))
    <data_root_2d>/<scene>/color/<frame>.jpg          320x240 RGB
    <data_root_2d>/<scene>/depth/<frame>.png          uint16 millimetres
    <data_root_2d>/<scene>/pose/<frame>.txt           4x4 camera->world
    <data_root_2d>/<scene>/label/<frame>.png          2D class ids (optional: training only)
    <caption_path>                                    {scene: {frame: caption}} (optional)

What is kept from the reference: colours back to 0..255 (``(feats + 1) * 127.5``, :137), the -100 / 255 -> ignore label remap
(:115-116), frames sorted numerically (:146-148), depth / 1000 (:170), the visibility filter 400 <= visible points <= val_keep
(:199-209), images resized to 512x512 (cv2.resize default = bilinear with half-pixel centres, :211; the 320x240 ScanNet frames
are UPSAMPLED, where PIL's bilinear filter has the same support).  The point -> pixel mapping with the depth-occlusion test
runs on the device (``ops.compute_mapping``, pinned by the reference's golden vectors).
Image decoding uses PIL (cv2 / imageio are not installed in this environment).  The ``.pth`` triple is loaded with
``weights_only=True`` (numpy arrays allow-listed), never unpickling arbitrary objects.
"""
from __future__ import annotations

import glob
import json
import os

import numpy as np
import torch

from . import synthetic


def _load_points(path):
    safe = []
    try:
        from numpy._core import multiarray as _ma
    except Exception:  # numpy < 2
        from numpy.core import multiarray as _ma
    safe += [_ma._reconstruct, np.ndarray, np.dtype]
    for name in ("Float32DType", "Float64DType", "Int64DType", "Int32DType", "UInt8DType", "Int16DType", "UInt16DType"):
        t = getattr(np.dtypes, name, None) if hasattr(np, "dtypes") else None
        if t is not None:
            safe.append(t)
    with torch.serialization.safe_globals(safe):
        obj = torch.load(path, weights_only=True)
    coords, feats, labels = obj[0], obj[1], obj[2]
    to_np = lambda a: a.numpy() if torch.is_tensor(a) else np.asarray(a)
    return to_np(coords), to_np(feats), to_np(labels)


def _resize_bilinear(img, out_h=512, out_w=512):
    """cv2.resize(img, (out_w, out_h)) with INTER_LINEAR semantics (half-pixel centres, edge clamp), float32 output"""
    h, w = img.shape[:2]
    ys = (np.arange(out_h) + 0.5) * h / out_h - 0.5
    xs = (np.arange(out_w) + 0.5) * w / out_w - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, h - 1)
    x0 = np.clip(np.floor(xs).astype(int), 0, w - 1)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    wy = np.clip(ys - np.floor(ys), 0, 1)[:, None, None] * (ys >= 0)[:, None, None]
    wx = np.clip(xs - np.floor(xs), 0, 1)[None, :, None] * (xs >= 0)[None, :, None]
    img = img.astype(np.float32)
    a, b, c, d = img[y0][:, x0], img[y0][:, x1], img[y1][:, x0], img[y1][:, x1]
    out = (a * (1 - wx) + b * wx) * (1 - wy) + (c * (1 - wx) + d * wx) * wy
    # the reference resizes the uint8 frame, so the result is a uint8 image again (cv2 rounds in 11-bit fixed point: it can
    # differ from this round-to-nearest by one grey level on exact .5 cases - parity unpinned, cv2 is not installed here)
    return np.clip(np.rint(out), 0, 255).astype(np.float32)


def load_scene(data_root, data_root_2d, scene_name, split="val", caption_path=None, ignore_label=20, device=None, val_keep=10000000,
               min_visible=400, ignore_categories=None, min_valid=10):
    """-> (synthetic.Scene with per-view depth maps, kept frame ids): a scene in the same container the synthetic generator
    fills, so SceneOnDevice / infer_scene work unchanged.  Views failing the reference's visibility filter are dropped.  With a
    `device` the point -> pixel mapping runs on it (xm3d_compute_mapping), else through the numpy restatement."""
    from PIL import Image

    coords, feats, labels = _load_points(os.path.join(data_root, split, scene_name + "_vh_clean_2.pth"))
    coords = np.asarray(coords, dtype=np.float64)
    colors = (np.asarray(feats, dtype=np.float64) + 1.0) * 127.5 if not (np.isscalar(feats) and feats == 0) else np.zeros_like(coords)
    labels = np.asarray(labels).astype(np.int64).copy()
    labels[(labels == -100) | (labels == 255)] = ignore_label
    ignore_cat = np.asarray([ignore_label] if ignore_categories is None else list(ignore_categories))
    captions = {}
    if caption_path and os.path.exists(caption_path):
        with open(caption_path) as f:
            captions = json.load(f).get(scene_name, {})
    scene_dir = os.path.join(data_root_2d, scene_name)
    frames = sorted(glob.glob(os.path.join(scene_dir, "color", "*")), key=lambda p: int(os.path.basename(p)[:-4]))
    if not frames:
        raise FileNotFoundError(f"no colour frames under {scene_dir}/color")
    sc = synthetic.Scene(coords, colors, labels)
    ids = []
    pts_dev = torch.from_numpy(coords).to(device) if device is not None else None
    for path in frames:
        fid = os.path.basename(path)[:-4]
        pose = np.loadtxt(os.path.join(scene_dir, "pose", fid + ".txt"))
        depth = np.asarray(Image.open(os.path.join(scene_dir, "depth", fid + ".png"))).astype(np.float64) / 1000.0
        if device is not None:
            from . import ops

            m = ops.compute_mapping(pts_dev, pose, synthetic.scannet_intrinsics(), depth=torch.from_numpy(depth).to(device)).cpu().numpy()
        else:
            m = synthetic.project_points(pose, coords, depth)
        nvis = int(m[:, 2].sum())
        n_valid = int((~np.isin(labels[m[:, 2] == 1], ignore_cat)).sum())
        if nvis < min_visible or n_valid < min_valid or nvis > val_keep:
            continue  # data_loader_infer.py:199-209
        img = np.asarray(Image.open(path).convert("RGB"))
        sc.poses.append(pose)
        sc.images.append(_resize_bilinear(img))
        sc.captions.append(captions.get(fid, ""))
        sc.depths.append(depth)
        ids.append(fid)
    return sc, ids
