"""Per-stage GPU time of the dense branch (B views, bf16): every stage is captured as its own HIP graph and the replay is
timed, so the numbers are launch-overhead free and add up to the graph time bench.py reports.
python tools/prof_stages.py [B] [cl]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cl = "cl" in sys.argv[2:]
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(0)
model = XMASK3d(cfg).eval().to(dev).set_dense_dtype(torch.bfloat16)
if cl:
    model.set_channels_last(True)
if "heads16" in sys.argv[2:]:
    model.cast_head_weights()
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
batch = pipeline.build_scene_batch(sd, [i % 5 for i in range(B)], vox, [np.diag([50.0, 50.0, 50.0, 1.0])] * B)


def graph_time(fn, reps=10):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            out = fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out, g


with torch.no_grad():
    _, cond, _ = model.encode_3d(batch["sinput"], batch["inds_reconstruct"], B)
    bb = model.backbone
    fe = bb.feature_extractor
    ext = fe.ldm_extractor
    keep = []
    images = model.normalize_images(batch["img"].float())
    img = bb.prepare(images)
    rows = []
    t, (latent, enc_feats), g = graph_time(lambda: ext.encode(img)); keep.append(g); rows.append(("VAE encoder", t))
    c, ce = fe.conditioning(cond)
    c, ce = c.to(latent.dtype), ce.to(latent.dtype)
    t, unet_feats, g = graph_time(lambda: ext.unet_taps(latent, c, ce)); keep.append(g); rows.append(("UNet (taps, pruned)", t))
    t, dec_feats, g = graph_time(lambda: ext.decode_taps(latent)); keep.append(g); rows.append(("VAE decoder (taps, pruned)", t))
    feats = [*enc_feats, *unet_feats, *dec_feats]

    def proj():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return bb.forward_features(feats, (512, 512))
    t, feature, g = graph_time(proj); keep.append(g); rows.append(("feature projections", t))
    low = model.low_precision_heads
    fin = {k: (v if getattr(model, "heads_native_bf16", False) else v.float()) for k, v in feature.items()}

    def pix():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=low):
            return model.sem_seg_head.pixel_decoder.forward_features(fin)
    t, (mf, _, ms), g = graph_time(pix); keep.append(g); rows.append(("pixel decoder (MSDeformAttn)", t))

    def pred():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=low):
            return model.sem_seg_head.predictor(ms, mf, None)
    t, outputs, g = graph_time(pred); keep.append(g); rows.append(("transformer decoder", t))
    for k in ("pred_masks", "mask_embed", "mask_pooled_features"):
        outputs[k] = outputs[k].float()
    outputs["images"] = batch["img"].float() / 255.0

    def cat():
        o = dict(outputs)
        o.update(model.category_head(o))
        return model.cal_pred_logits(o)
    t, _, g = graph_time(cat); keep.append(g); rows.append(("category head + logits", t))
    t, _, g = graph_time(lambda: model.clip_head(outputs["images"], outputs["pred_masks"])); keep.append(g); rows.append(("mask-CLIP", t))
    t, _, g = graph_time(lambda: model.dense_forward(batch["img"], cond)); keep.append(g); rows.append(("dense_forward (one graph, serial)", t))
    tot = sum(r[1] for r in rows[:-1])
    print(f"B={B} channels_last={cl}")
    for n, t in rows:
        print(f"  {n:38s} {t:8.2f} ms  {100 * t / tot:5.1f}%")
    print(f"  {'sum of stages':38s} {tot:8.2f} ms")
