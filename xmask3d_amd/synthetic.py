"""Synthetic ScanNet-shaped inputs and the point->pixel mapping.

Host-side data preparation (numpy), the counterpart of the reference's dataset
classes for a world without ScanNet on disk:

* pinhole projection with frustum / boundary / optional depth-occlusion test:
  same contract as ``PointCloudToImageMapper.compute_mapping``
  (/root/reference/models/utils/fusion_util.py:46-142) with the fixed ScanNet
  intrinsics of ``getMapping`` (/root/reference/models/utils/mapping_util.py:10-39)
* per-view sample assembly as ``ScannetLoaderFull.__getitem__`` does it
  (/root/reference/dataset/data_loader_infer.py:161-283): visible subset ->
  voxelise -> coords/feats/inds_reconstruct/x_label/y_label.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

IMAGE_DIM = (320, 240)  # (width, height) of the mask grid
CUT_BOUND = 10
VIS_THRES = 0.25


def scannet_intrinsics() -> np.ndarray:
    """577.870605 px focal at 640x480 rescaled to 320x240 (mapping_util.py:17-31)."""
    fx = fy = 577.870605
    mx, my = 319.5, 239.5
    src_w, src_h = 640, 480
    dst_w, dst_h = IMAGE_DIM
    k = np.eye(4)
    resize_w = int(math.floor(dst_h * float(src_w) / float(src_h)))
    k[0, 0] = fx * float(resize_w) / float(src_w)
    k[1, 1] = fy * float(dst_h) / float(src_h)
    k[0, 2] = mx * float(dst_w - 1) / float(src_w - 1)
    k[1, 2] = my * float(dst_h - 1) / float(src_h - 1)
    return k


def project_points(camera_to_world, pts, depth=None, intrinsic=None, image_dim=IMAGE_DIM, cut_bound=CUT_BOUND,
                   vis_thres=VIS_THRES):
    """(N,3) int array [row, col, visible] with rows/cols zeroed for invisible points."""
    k = scannet_intrinsics() if intrinsic is None else intrinsic
    n = pts.shape[0]
    homo = np.concatenate([pts, np.ones((n, 1))], axis=1).T
    cam = np.linalg.inv(camera_to_world) @ homo
    z = cam[2].copy()
    z[np.abs(z) < 1e-8] = 1.0
    u = np.round(cam[0] * k[0][0] / z + k[0][2]).astype(int)
    v = np.round(cam[1] * k[1][1] / z + k[1][2]).astype(int)
    vis = (cam[2] > 0) & (u >= cut_bound) & (v >= cut_bound) & (u < image_dim[0] - cut_bound) & (v < image_dim[1] - cut_bound)
    if depth is not None and vis.any():
        cand = np.nonzero(vis)[0]
        ok = (v[cand] >= 0) & (v[cand] < depth.shape[0]) & (u[cand] >= 0) & (u[cand] < depth.shape[1])
        cand = cand[ok]
        d = depth[v[cand], u[cand]]
        keep = np.abs(d - cam[2][cand]) <= vis_thres * d
        vis = np.zeros_like(vis)
        vis[cand[keep]] = True
    out = np.zeros((n, 3), dtype=int)
    out[vis, 0] = v[vis]
    out[vis, 1] = u[vis]
    out[vis, 2] = 1
    return out


def _sample_rect(rng, n, origin, e1, e2):
    a, b = rng.uniform(size=(2, n))
    return origin + a[:, None] * e1 + b[:, None] * e2


def make_box_room(rng, n_points, size, n_boxes):
    """Area-proportional samples on floor + 4 walls of a room plus random boxes, 3 mm jitter."""
    sx, sy, sz = size
    rects = [
        (np.zeros(3), np.array([sx, 0, 0.0]), np.array([0, sy, 0.0])),
        (np.zeros(3), np.array([sx, 0, 0.0]), np.array([0, 0, sz])),
        (np.array([0, sy, 0.0]), np.array([sx, 0, 0.0]), np.array([0, 0, sz])),
        (np.zeros(3), np.array([0, sy, 0.0]), np.array([0, 0, sz])),
        (np.array([sx, 0, 0.0]), np.array([0, sy, 0.0]), np.array([0, 0, sz])),
    ]
    for _ in range(n_boxes):
        w, d = rng.uniform(0.4, 1.6, size=2)
        h = rng.uniform(0.4, 1.2)
        ox = rng.uniform(0.1, max(0.11, sx - w - 0.1))
        oy = rng.uniform(0.1, max(0.11, sy - d - 0.1))
        o = np.array([ox, oy, 0.0])
        ex, ey, ez = np.array([w, 0, 0.0]), np.array([0, d, 0.0]), np.array([0, 0, h])
        rects += [(o + ez, ex, ey), (o, ex, ez), (o + ey, ex, ez), (o, ey, ez), (o + ex, ey, ez)]
    areas = np.array([np.linalg.norm(np.cross(e1, e2)) for _, e1, e2 in rects])
    counts = np.floor(areas / areas.sum() * n_points).astype(int)
    pts = np.concatenate([_sample_rect(rng, c, o, e1, e2) for c, (o, e1, e2) in zip(counts, rects) if c > 0])
    pts += rng.normal(0, 0.003, size=pts.shape)
    return pts


def camera_pose(position, yaw_deg, pitch_deg=0.0):
    """camera->world; camera looks along +z_cam, x right, y down (ScanNet convention)."""
    yaw, pitch = math.radians(yaw_deg), math.radians(pitch_deg)
    fwd = np.array([math.cos(yaw) * math.cos(pitch), math.sin(yaw) * math.cos(pitch), math.sin(pitch)])
    right = np.array([math.sin(yaw), -math.cos(yaw), 0.0])
    down = np.cross(fwd, right)
    pose = np.eye(4)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, fwd, position
    return pose


def noise_image(rng, h=240, w=320, out=512):
    """low-pass filtered uniform noise, bilinear resize to out x out, HWC float32 in 0..255."""
    small = rng.uniform(0, 255, size=(h // 8, w // 8, 3))
    ys = (np.arange(out) + 0.5) * small.shape[0] / out - 0.5
    xs = (np.arange(out) + 0.5) * small.shape[1] / out - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, small.shape[0] - 2)
    x0 = np.clip(np.floor(xs).astype(int), 0, small.shape[1] - 2)
    wy = np.clip(ys - y0, 0, 1)[:, None, None]
    wx = np.clip(xs - x0, 0, 1)[None, :, None]
    a, b = small[y0][:, x0], small[y0][:, x0 + 1]
    c, d = small[y0 + 1][:, x0], small[y0 + 1][:, x0 + 1]
    return ((a * (1 - wx) + b * wx) * (1 - wy) + (c * (1 - wx) + d * wx) * wy).astype(np.float32)


@dataclass
class Scene:
    points: np.ndarray  # (N,3) f64 metres
    colors: np.ndarray  # (N,3) f64 0..255
    labels: np.ndarray  # (N,) int
    poses: list = field(default_factory=list)  # camera->world 4x4 per view
    images: list = field(default_factory=list)  # (512,512,3) f32 0..255 per view
    captions: list = field(default_factory=list)
    depths: list = field(default_factory=list)  # optional (240,320) f64 metres per view: occlusion test of the mapping


def scene_s0(seed=5557) -> Scene:
    rng = np.random.RandomState(seed)
    pts = make_box_room(rng, 8192, (3.0, 3.0, 2.5), 0)
    # a camera at the box centre sees one wall; add its ceiling-free interior only
    n = pts.shape[0]
    return Scene(pts, rng.randint(0, 256, size=(n, 3)).astype(np.float64), rng.randint(0, 15, size=n),
                 [camera_pose((1.5, 1.5, 1.25), 0.0)], [noise_image(rng)], ["a room"])


def scene_s1(seed=5557, n_points=120000, n_views=5) -> Scene:
    rng = np.random.RandomState(seed)
    pts = make_box_room(rng, n_points, (6.0, 5.0, 2.6), 12)
    n = pts.shape[0]
    poses = [camera_pose((3.0, 2.5, 1.5), 360.0 / n_views * v) for v in range(n_views)]
    images = [noise_image(rng) for _ in range(n_views)]
    return Scene(pts, rng.randint(0, 256, size=(n, 3)).astype(np.float64), rng.randint(0, 15, size=n), poses, images,
                 ["a room"] * n_views)


def view_subset(scene: Scene, view: int, depth=None):
    """Visible points of one view + their pixel rows/cols (data_loader_infer.py:161-176,255-258)."""
    if depth is None and len(scene.depths) > view:
        depth = scene.depths[view]
    m = project_points(scene.poses[view], scene.points, depth)
    vis = m[:, 2] == 1
    rows = m[vis]
    keep = np.all(rows != 0, axis=1)  # the reference drops rows containing any zero from x/y labels
    return vis, rows[keep, 0], rows[keep, 1]
