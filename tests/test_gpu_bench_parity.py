"""Parity of the configurations bench.py measures against the CPU oracle (oracle/model_oracle.forward_cpu), stage by stage
down to the per-point logits ``logit_scale * norm(fused) @ norm(text).T`` (/root/reference/run/infer.py:556-558) and the
scene votes.

* fp32 (the reference's arithmetic): per-point logits within north_star's 1e-3 (absolute, logits are ``scale * cos`` with
  scale ~14) on every point whose discrete mask ownership agrees between the two runs (a flipped 0.5-threshold moves a point
  to another mask embedding: that is a discrete event, counted separately and bounded).
* bf16 + channels-last + bf16 head weights + 3 HIP graphs + 2 scenes (10 views) per forward = the bench default (with 4
  scenes): measured budgets per stage, asserted below and quoted in DESIGN.md §5.
The dense nets of the oracle are the model's own torch modules on the CPU in fp32 (PARITY UNPINNED for their numerics: no
reference fixture exists); voxelisation, MSDeformAttn and the fusion loop of the oracle are pinned by reference goldens.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

T50 = np.diag([50.0, 50.0, 50.0, 1.0])


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-20)


@pytest.fixture(scope="module")
def setup(dev):
    from xmask3d_amd import synthetic
    from xmask3d_amd.config import load_cfg_from_cfg_file
    from xmask3d_amd.xmask3d import XMASK3d

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_cfg_from_cfg_file(os.path.join(root, "configs", "xmask3d_scannet_B15N4.yaml"))
    torch.manual_seed(cfg.manual_seed)
    cpu = XMASK3d(cfg).eval()
    scenes = [synthetic.scene_s1(seed=5557), synthetic.scene_s1(seed=5558)]
    return cfg, cpu, scenes


_ORACLE_CACHE = {}


def oracle_view(cpu, scene, v, key):
    """one view through the CPU oracle, all Q mask rows kept (so that the ownership of every point can be compared)"""
    from oracle import model_oracle, voxel_oracle
    from xmask3d_amd import synthetic

    if key in _ORACLE_CACHE:
        return _ORACLE_CACHE[key]
    vis, rows, cols = synthetic.view_subset(scene, v)
    pts = scene.points[vis]
    grid, inds, inv = voxel_oracle.voxelize_with_matrix(pts, T50)
    coords = torch.from_numpy(np.concatenate([np.zeros((len(grid), 1)), grid], 1).astype(np.int32))
    feats = torch.from_numpy((scene.colors[vis][inds] / 127.5 - 1).astype(np.float32))
    cbatch = {"sinput": model_oracle.CpuSparseTensor(feats, coords), "img": torch.from_numpy(scene.images[v]).permute(2, 0, 1)[None],
              "x_label": torch.from_numpy(rows).long(), "y_label": torch.from_numpy(cols).long(),
              "inds_reconstruct": torch.from_numpy(inv), "captions": ("a room",),
              "ori_coords": torch.cat([torch.zeros(len(pts), 1), torch.from_numpy(pts).float()], 1),
              "point_offsets": [0, len(pts)], "compact_outputs": False}
    _, ref = model_oracle.forward_cpu(cpu, cbatch)
    _ORACLE_CACHE[key] = ref
    return ref


def point_logits(fused, outputs):
    text = F.normalize(outputs["text_embed"].float(), dim=-1)
    return outputs["logit_scale"].float() * (F.normalize(fused.float(), dim=-1) @ text.t())


def stage_report(out, b, sel, ref):
    """errors of batch entry b of a device forward (`sel` = its point range) against the oracle outputs of that view"""
    rep = {k: _rel(out[k][b], ref[k][0]) for k in ("pred_masks", "mask_embed")}
    # mask-CLIP thresholds every mask per 14x14 patch (clip.py:272-310): a mask logit within rounding of the threshold flips a
    # patch of ONE query's attention mask - a discrete event.  Bound the queries that did not flip, count the ones that did.
    ce = (out["mask_embed_clip"][b].float().cpu() - ref["mask_embed_clip"][0]).abs().amax(-1) / ref["mask_embed_clip"][0].abs().max()
    rep["mask_embed_clip"] = ce.sort().values[: ce.numel() - 2].max().item()      # all but the two worst queries
    rep["clip_queries_flipped"] = float((ce > 10 * max(rep["mask_embed_clip"], 1e-6)).sum())
    rep["pred_3d"] = _rel(out["pred_3d"][sel], ref["pred_3d"])
    rep["pred_logits_abs"] = (out["pred_logits"][b].float().cpu() - ref["pred_logits"][0]).abs().max().item()
    m_g, m_r = out["final_mask_3d"][b].cpu(), ref["final_mask_3d"][0]          # (Q, Np) bool, all Q rows on both sides
    agree = (m_g == m_r).all(0)
    rep["ownership_agree"] = agree.float().mean().item()
    fg, fr = out["fused_pred_feature"][b].float().cpu(), ref["fused_pred_feature"][0]
    rep["fused_rel"] = ((fg - fr)[agree].abs().max() / fr.abs().max()).item()
    lg, lr = point_logits(fg, {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in out.items() if k in ("text_embed", "logit_scale")}), \
        point_logits(fr, ref)
    rep["point_logits_abs"] = (lg - lr)[agree].abs().max().item()
    rep["point_label_agree"] = (lg.argmax(1) == lr.argmax(1))[agree].float().mean().item()
    rep["binary_agree"] = (out["binary_pred"][sel].cpu() == ref["binary_pred"]).float().mean().item()
    return rep


def _forward_group(model, sds, vox):
    from xmask3d_amd import pipeline

    batch = pipeline.build_group_batch([(sd, list(range(len(sd.views)))) for sd in sds], vox, [[T50] * len(sd.views) for sd in sds])
    batch["compact_outputs"] = False
    with torch.no_grad():
        front = model.eval_front(batch)
        out = model.eval_fuse(batch, front, model.eval_dense(batch, front))
    return batch, out


# fp32 bounds (relative to the tensor's max magnitude unless "_abs").  Measured over several boxes (gpurun_out/r2_parity*.log,
# DESIGN.md §5): pred_3d 5e-6, pred_masks 2e-4, mask_embed 4e-4, mask_embed_clip 2e-6, pred_logits 3e-4, fused 5e-6..2e-4,
# per-point logits 8e-6..1.5e-4 - eager and graph replay alike.  The fused feature of a point carries the error of the mask
# embedding of the query that owns it (the 4e-4 sits in one or two queries per view, which own points on some runs and none
# on others), so its bound is the embedding's.  north_star: per-point logits within 1e-3.
FP32 = {"pred_3d": 5e-5, "pred_masks": 1e-3, "mask_embed": 1e-3, "mask_embed_clip": 1e-4, "pred_logits_abs": 1e-3,
        "fused_rel": 1e-3, "point_logits_abs": 1e-3}
# bench configuration (bf16 frozen nets + bf16 head GEMMs).  Measured: pred_masks 5.1e-2, mask_embed 6.3e-2, mask_embed_clip
# 3.9e-2, pred_logits 4.2e-2, fused 2.0e-2, per-point logits 1.8e-2 (scale*cos, scale ~14), ownership 98.3 %, labels 100 %.
# With the HIP flash attention in the path: 4.4e-2 / 6.8e-2 / 3.1e-2 / 4.1e-2 / 3.6e-2 / 2.4e-2, ownership 96.7 %.
# With the fused bf16 GroupNorm / LayerNorm / FPN kernels of round 2's last build: 4.4e-2 / 6.1e-2 / 3.3e-2 / 3.9e-2 / 2.4e-2 /
# 1.7e-2, ownership 95.0-98.5 %, labels 100 %.
BF16 = {"pred_3d": 4e-2, "pred_masks": 8e-2, "mask_embed": 1e-1, "mask_embed_clip": 7e-2, "pred_logits_abs": 8e-2,
        "fused_rel": 6e-2, "point_logits_abs": 5e-2}


# (fp32 eager batch 1, the reference's own configuration: tests/test_gpu_model.py::test_eval_forward_matches_cpu_oracle, same bounds)
# fp32 with the OPT-IN split-operand convolutions (XM3D_CONV_F32=hip, ops.conv3x3_f32: 2e-5 per layer): the ~60 convolutions in
# sequence bring the stages to ~1e-3 (measured: pred_masks 1.0e-3, mask_embed 1.5e-3, pred_logits 8e-4, fused 1.2e-3, per-point
# logits 6e-4).  Inside north_star's 1e-3 on the per-point logits on these views, but without the 6x margin of the exact-f32
# convolutions - the reason it is not the default of the fp32 configuration.


@pytest.mark.parametrize("mode", ["fp32_graph_nhwc", "bf16_bench", "fp32_library"])
def test_configuration_matches_oracle_per_stage(dev, setup, mode, monkeypatch):
    from xmask3d_amd import pipeline

    cfg, cpu, scenes = setup
    # fp32_graph_nhwc = the fp32 configuration as shipped (every convolution / GEMM of the frozen nets: two-term split in IEEE halves on the
    # matrix cores, three passes, ~1e-6 per layer); fp32_library = the same with the library's f32 convolutions and GEMMs (same bounds)
    if mode == "fp32_library":
        monkeypatch.setenv("XM3D_CONV_F32", "library")
        monkeypatch.setenv("XM3D_GEMM_F32", "library")
    dtype = torch.bfloat16 if mode == "bf16_bench" else torch.float32
    model = pipeline.make_inference_model(cpu, dev, dtype, channels_last=True, graphs=True)
    # bf16: 2 scenes x 5 views in ONE forward (batch 10), as bench does; fp32: one scene per forward, like bench's fp32 leg
    # (the shipped MIOpen find-db holds the fp32 NHWC shapes at batch 5; untuned shapes cost minutes of solver search)
    sds = [pipeline.SceneOnDevice(sc, dev) for sc in (scenes if mode == "bf16_bench" else scenes[:1])]
    vox = pipeline.default_voxelizer(cfg.voxel_size, dev)
    batch, out = _forward_group(model, sds, vox)
    off = batch["point_offsets"]
    bounds = BF16 if mode == "bf16_bench" else FP32
    worst = {}
    for (si, v) in (((0, 0), (1, 2)) if mode == "bf16_bench" else ((0, 0), (0, 3))):   # through the CPU oracle (20 - 40 s of host time each)
        b = si * 5 + v
        ref = oracle_view(cpu, scenes[si], v, (si, v))
        rep = stage_report(out, b, slice(off[b], off[b + 1]), ref)
        print(f"[parity {mode} scene {si} view {v}] " + " ".join(f"{k}={x:.3e}" for k, x in rep.items()))
        for k, x in rep.items():
            worst[k] = max(worst.get(k, 0.0), x) if not k.endswith("agree") else min(worst.get(k, 1.0), x)
    assert worst["clip_queries_flipped"] <= 2
    for k, bound in bounds.items():
        assert worst[k] <= bound, f"{mode}: {k} = {worst[k]:.3e} > {bound:.1e}"
    assert worst["binary_agree"] > 0.999
    # 0.5-threshold flips under the bf16 budget: with random weights many mask logits sit near the threshold; measured
    # 95.0-98.5 % per view across runs, with 100 % of the per-point LABELS unchanged (a flipped point moves between masks of
    # the same class)
    assert worst["ownership_agree"] > (0.92 if mode == "bf16_bench" else 0.995)
    assert worst["point_label_agree"] > (0.97 if mode == "bf16_bench" else 0.9995)


def test_bench_configuration_votes_match_fp32_reference_path(dev, setup):
    """scene votes of the bench configuration (bf16, NHWC, graphs, 2 scenes per forward, batched fusion + post-processing)
    against the fp32 eager batch-1 per-view loop (the reference driver's structure, run/infer.py:428-694)"""
    from xmask3d_amd import pipeline

    cfg, cpu, scenes = setup
    fast = pipeline.make_inference_model(cpu, dev, torch.bfloat16)
    slow = pipeline.make_inference_model(cpu, dev, torch.float32, channels_last=False, graphs=False)
    sds = [pipeline.SceneOnDevice(sc, dev) for sc in scenes]
    vox = pipeline.default_voxelizer(cfg.voxel_size, dev)
    M = [[T50] * 5, [T50] * 5]
    res = pipeline.infer_scenes(fast, sds, cfg, vox, M)
    for sd, mats, got in zip(sds, M, res):
        want = pipeline.infer_scene(slow, sd, cfg, vox, mats, views_per_batch=1)
        for name, a, b in zip(("fused", "2d", "3d"), got, want):
            agree = (a == b).float().mean().item()
            print(f"[votes {name}] agreement {agree:.4f}")
            # every label carries the bf16 budget of the bench configuration: the sparse nets run the plain-bf16 form since round 4
            # (measured 3D 0.9996 - 0.9999; with XM3D_SPARSE=f32 the 3D-only labels agree to 0.9999+), fused / 2D also the dense branch's
            assert agree > (0.998 if name == "3d" else 0.97), (name, agree)


@pytest.mark.parametrize("name", ["xmask3d_scannet_B12N7", "xmask3d_scannet_B170N30"])
def test_other_benchmark_configs_match_oracle_per_stage(dev, name):
    """BASELINE.json configs 4 and 5 (12 base / 7 novel classes; the 200-class ScanNet200 head, 170 / 30) through the same
    per-stage comparison against the CPU oracle as B15N4 above, in the fp32 configuration (north_star's 1e-3 on the per-point
    logits) - not only shapes and ranges."""
    from xmask3d_amd import pipeline, synthetic
    from xmask3d_amd.config import load_cfg_from_cfg_file
    from xmask3d_amd.xmask3d import XMASK3d

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_cfg_from_cfg_file(os.path.join(root, "configs", name + ".yaml"))
    torch.manual_seed(cfg.manual_seed)
    cpu = XMASK3d(cfg).eval()
    scene = synthetic.scene_s1(seed=5557)
    model = pipeline.make_inference_model(cpu, dev, torch.float32, channels_last=True, graphs=True)
    vox = pipeline.default_voxelizer(cfg.voxel_size, dev)
    batch, out = _forward_group(model, [pipeline.SceneOnDevice(scene, dev)], vox)
    off = batch["point_offsets"]
    n_test = len(cfg.category_split["base_category"]) + len(cfg.category_split["novel_category"])
    assert out["pred_logits"].shape[-1] == n_test + 1
    for v in (4,):  # one view per config: the CPU oracle forward is 35 s of host time per view
        ref = oracle_view(cpu, scene, v, (name, v))
        rep = stage_report(out, v, slice(off[v], off[v + 1]), ref)
        print(f"[parity {name} fp32 view {v}] " + " ".join(f"{k}={x:.3e}" for k, x in rep.items()))
        assert rep["clip_queries_flipped"] <= 2
        for k, bound in FP32.items():
            assert rep[k] <= bound, f"{name}: {k} = {rep[k]:.3e} > {bound:.1e}"
        assert rep["binary_agree"] > 0.999 and rep["ownership_agree"] > 0.995 and rep["point_label_agree"] > 0.9995
    if name.endswith("B170N30"):
        # the bench (bf16) configuration on the 200-class head, against the SAME oracle view: the measured bf16 budgets of B15N4 hold
        # (the class count only widens the text matrix of the last product)
        del model
        fast = pipeline.make_inference_model(cpu, dev, torch.bfloat16, channels_last=True, graphs=True)
        batch, out = _forward_group(fast, [pipeline.SceneOnDevice(scene, dev)], vox)
        off = batch["point_offsets"]
        rep = stage_report(out, 4, slice(off[4], off[5]), oracle_view(cpu, scene, 4, (name, 4)))
        print(f"[parity {name} bf16 view 4] " + " ".join(f"{k}={x:.3e}" for k, x in rep.items()))
        assert rep["clip_queries_flipped"] <= 2
        for k, bound in BF16.items():
            assert rep[k] <= bound, f"{name} bf16: {k} = {rep[k]:.3e} > {bound:.1e}"
        # 200 classes with random text rows: neighbouring classes sit closer than with 19, so more labels flip inside the same budget
        assert rep["binary_agree"] > 0.999 and rep["ownership_agree"] > 0.92 and rep["point_label_agree"] > 0.90
