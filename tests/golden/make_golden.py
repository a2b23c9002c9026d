"""Generate golden vectors by IMPORTING the reference (build container only).

Run:  python tests/golden/make_golden.py          (needs /root/reference)
Outputs small .npz files next to this script.  Only inputs and expected outputs
are stored - no reference source text.  /root/reference never travels to the GPU
box; the committed .npz files do.

What is captured (reference file -> fixture):
  dataset/voxelization_utils.py, dataset/voxelizer.py      -> voxel_*.npz
  .../pixel_decoder/ops/functions/ms_deform_attn_func.py    -> msda_*.npz
      (ms_deform_attn_core_pytorch = the reference's own CPU path, plus autograd grads)
  .../transformer_decoder/position_encoding.py              -> sine_pe.npz
  models/utils/fuser.py, models/modeling/meta_arch/helper.py-> fuser.npz, ensemble.npz
  models/utils/fusion_util.py, mapping_util.py              -> mapping.npz
  models/modeling/diffusion/gaussian_diffusion.py           -> diffusion.npz
  util/config.py + config/scannet/*.yaml                    -> config_b15n4.json
  models/modeling/meta_arch/{mink_unet,resnet_base,pc_processor}.py
      built through xmask3d_amd.me_compat.install_as_minkowski_engine()   -> seam_minkunet_keys.json
      (state_dict key -> shape of PC_Processor(34C) / PC_Binary_Processor(18A))
      and RUN (the reference's own forward code) on oracle/me_cpu_stub.py -> seam_minkunet_forward.npz
  .../pixel_decoder/ops/functions/ms_deform_attn_func.py + ops/modules/ms_deform_attn.py
      imported with xmask3d_amd.msda.install_as_msda(): the positional calls the reference's
      MSDeformAttnFunction makes into the extension module, recorded -> seam_msda_call.npz / .json
"""
import collections
import collections.abc
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

# python>=3.10 removed these aliases which the reference still uses
collections.Sequence = collections.abc.Sequence
collections.Iterable = collections.abc.Iterable
sys.path.insert(0, REF)
sys.dont_write_bytecode = True


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()})


def gen_voxel():
    from dataset.voxelization_utils import fnv_hash_vec, sparse_quantize
    from dataset.voxelizer import Voxelizer

    kat_in = np.array([[0, 0, 0], [1, 2, 3], [287, 130, 209], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float64)
    kat_keys = fnv_hash_vec(kat_in)
    sq_in = np.array([[1, 0, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [5, 5, 5]], dtype=np.float64)
    sq_inds, sq_inv = sparse_quantize(sq_in, return_index=True)

    rng = np.random.RandomState(20240607)
    big = np.floor(rng.uniform(0, 400, size=(20000, 3)))
    big[5000:9000] = big[1000:5000]  # forced duplicates
    big_keys = fnv_hash_vec(big)
    big_inds, big_inv = sparse_quantize(big, return_index=True)
    save("voxel_kat.npz", kat_in=kat_in, kat_keys=kat_keys, sq_in=sq_in, sq_inds=sq_inds, sq_inv=sq_inv,
         big=big, big_keys=big_keys, big_inds=big_inds, big_inv=big_inv)

    # full voxelize() with the loaders' configuration (dataset/point_loader.py:52-60,100-107)
    vox = Voxelizer(
        voxel_size=0.02, clip_bound=None, use_augmentation=True,
        scale_augmentation_bound=(0.9, 1.1),
        rotation_augmentation_bound=((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi)),
        translation_augmentation_ratio_bound=((-0.2, 0.2), (-0.2, 0.2), (0, 0)),
    )
    for tag, n, seed in (("a", 8192, 5557), ("b", 30000, 7)):
        r = np.random.RandomState(seed)
        # points on box faces with mm jitter: many points share a voxel
        pts = r.uniform(0, 1, size=(n, 3)) * np.array([3.0, 3.0, 2.5])
        face = r.randint(0, 3, size=n)
        pts[np.arange(n), face] = np.where(r.rand(n) < 0.5, 0.0, np.array([3.0, 3.0, 2.5])[face])
        pts += r.normal(0, 0.003, size=pts.shape)
        feats = r.randint(0, 256, size=(n, 3)).astype(np.float64)
        labels = r.randint(0, 15, size=n)
        np.random.seed(seed)
        state_probe = np.random.get_state()[1][:4].copy()
        np.random.seed(seed)
        M_v, M_r = None, None
        # capture the matrix by replaying the same seed on the reference's own method
        M_v, M_r = vox.get_transformation_matrix()
        np.random.seed(seed)
        locs, f2, l2, inv, inds = vox.voxelize(pts, feats.copy(), labels.copy(), return_ind=True)
        save(f"voxel_scene_{tag}.npz", pts=pts, feats=feats, labels=labels, seed=np.int64(seed),
             matrix=(M_r @ M_v), M_v=M_v, M_r=M_r, locs=locs, vfeats=f2, vlabels=l2, inv=inv, inds=inds,
             state_probe=state_probe)


def gen_msda():
    sys.path.insert(0, os.path.join(REF, "third_party/Mask2Former/mask2former/modeling/pixel_decoder"))
    from ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch

    def run(name, N, M, D, Lq, shapes, P, seed, dtype):
        torch.manual_seed(seed)
        shapes_t = torch.as_tensor(shapes, dtype=torch.long)
        L = len(shapes)
        S = int(shapes_t.prod(1).sum())
        lsi = torch.cat((shapes_t.new_zeros((1,)), shapes_t.prod(1).cumsum(0)[:-1]))
        value = (torch.rand(N, S, M, D) * 0.01).to(dtype)
        # locations deliberately spill outside [0,1] to exercise zero padding
        loc = (torch.rand(N, Lq, M, L, P, 2) * 1.3 - 0.15).to(dtype)
        w = (torch.rand(N, Lq, M, L, P) + 1e-5).to(dtype)
        w = w / w.sum(-1, keepdim=True).sum(-2, keepdim=True)
        value.requires_grad_(True); loc.requires_grad_(True); w.requires_grad_(True)
        out = ms_deform_attn_core_pytorch(value, shapes_t, loc, w)
        go = torch.randn_like(out)
        out.backward(go)
        save(name, value=value.detach().numpy(), shapes=shapes_t.numpy(), level_start=lsi.numpy(),
             loc=loc.detach().numpy(), w=w.detach().numpy(), out=out.detach().numpy(), grad_out=go.numpy(),
             g_value=value.grad.numpy(), g_loc=loc.grad.numpy(), g_w=w.grad.numpy())

    # the toy shape of ops/test.py:21-31 (seed 3), in f64 and f32
    run("msda_toy_f64.npz", 1, 2, 2, 2, [(6, 4), (3, 2)], 2, 3, torch.float64)
    run("msda_toy_f32.npz", 1, 2, 2, 2, [(6, 4), (3, 2)], 2, 3, torch.float32)
    # head dim 32 as used by the pixel decoder, reduced query count to keep the file small
    run("msda_d32_f32.npz", 2, 8, 32, 300, [(16, 16), (8, 8), (4, 4)], 4, 11, torch.float32)
    run("msda_d32_f64.npz", 1, 8, 32, 64, [(8, 12), (4, 6), (2, 3)], 4, 12, torch.float64)

    sys.path.insert(0, os.path.join(REF, "third_party/Mask2Former/mask2former/modeling/transformer_decoder"))
    from position_encoding import PositionEmbeddingSine

    pe = PositionEmbeddingSine(128, normalize=True)
    x = torch.zeros(2, 256, 6, 9)
    save("sine_pe.npz", pe=pe(x).numpy(), shape=np.array(x.shape))


def gen_fuser():
    from models.utils.fuser import FeatureMerger, mask_mapper
    from models.modeling.meta_arch.helper import ensemble_logits_with_labels

    torch.manual_seed(17)
    C = 24
    fuser = FeatureMerger(C)
    fc1 = torch.nn.Identity(); fc2 = torch.nn.Identity()

    class Cfg:  # the only attribute mask_mapper reads
        caption_contra_2d_pre = True

    Q, H, W = 7, 24, 32
    xs, ys, masks, embeds, p3d = [], [], [], [], []
    for n in (50, 80):
        xs.append(torch.randint(0, H, (n,))); ys.append(torch.randint(0, W, (n,)))
        m = torch.rand(Q, H, W); m[2] = 0.0  # an empty query
        masks.append(m); embeds.append(torch.randn(Q, C)); p3d.append(torch.randn(n, C))
    with torch.no_grad():
        out, out2d, out3d, out2dpre = mask_mapper(xs, ys, masks, embeds, p3d, fuser, fc1, fc2, Cfg)
    arrs = dict(W=fuser.linear.weight.detach().numpy(), b=fuser.linear.bias.detach().numpy())
    for i in range(2):
        arrs.update({f"x{i}": xs[i].numpy(), f"y{i}": ys[i].numpy(), f"mask{i}": masks[i].numpy(),
                     f"emb{i}": embeds[i].numpy(), f"p3d{i}": p3d[i].numpy(), f"fused{i}": out[i].numpy(),
                     f"f2d{i}": out2d[i].numpy(), f"f3d{i}": out3d[i].numpy(), f"f2dpre{i}": out2dpre[i].numpy()})
    save("fuser.npz", **arrs)

    logits = torch.randn(3, 5, 7)
    labels = [["a"], ["b", "c", "d"], ["e"], ["f", "g"]]
    save("ensemble.npz", logits=logits.numpy(), lens=np.array([len(l) for l in labels]),
         out_max=ensemble_logits_with_labels(logits, labels, "max").numpy(),
         out_mean=ensemble_logits_with_labels(logits, labels, "mean").numpy())


def gen_mapping():
    from models.utils.mapping_util import getMapping

    mapper = getMapping()
    r = np.random.RandomState(3)
    pts = r.uniform(-3, 3, size=(5000, 3))
    ang = 0.7
    pose = np.eye(4)
    pose[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]) @ np.array(
        [[1, 0, 0], [0, 0, 1], [0, -1, 0]])
    pose[:3, 3] = [0.2, -0.1, 0.3]
    m0 = mapper.compute_mapping(pose, pts, None)
    depth = r.uniform(0.5, 4.0, size=(240, 320))
    m1 = mapper.compute_mapping(pose, pts, depth)
    save("mapping.npz", pts=pts, pose=pose, depth=depth, map_nodepth=m0, map_depth=m1,
         intrinsic=mapper.intrinsics, image_dim=np.array(mapper.image_dim))


def gen_diffusion():
    from models.modeling.diffusion.gaussian_diffusion import get_named_beta_schedule

    betas = get_named_beta_schedule("ldm_linear", 1000)
    ac = np.cumprod(1.0 - betas)
    save("diffusion.npz", betas_head=betas[:4], sqrt_ac0=np.sqrt(ac[0]), sqrt_1m_ac0=np.sqrt(1 - ac[0]))


def gen_config():
    from util import config as refcfg

    cfg = refcfg.load_cfg_from_cfg_file(os.path.join(REF, "config/scannet/xmask3d_scannet_B15N4.yaml"))
    cfg = refcfg.merge_cfg_from_list(cfg, ["save_path", "out/x", "batch_size", "8", "train_gpu", "[0,1]"])

    def plain(o):
        if isinstance(o, dict):
            return {k: plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [plain(v) for v in o]
        return o

    with open(os.path.join(HERE, "config_b15n4.json"), "w") as f:
        json.dump(plain(dict(cfg)), f, indent=1, sort_keys=True)
    print("wrote config_b15n4.json")


def _load_ref_meta_arch(tag):
    """import the reference's mink_unet / resnet_base / pc_processor files under a private package name (their package
    __init__ chain needs detectron2); whatever module is registered as ``MinkowskiEngine`` at this moment is what they bind"""
    import importlib.util
    import types

    base = os.path.join(REF, "models/modeling/meta_arch")
    pkg = types.ModuleType(tag)
    pkg.__path__ = [base]
    sys.modules[tag] = pkg
    mods = {}
    for name in ("resnet_base", "mink_unet", "pc_processor"):
        spec = importlib.util.spec_from_file_location(f"{tag}.{name}", os.path.join(base, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        m.__package__ = tag
        sys.modules[f"{tag}.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def gen_seams():
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import me_cpu_stub
    from xmask3d_amd import me_compat, msda

    # (1) the reference's model files built on the PRODUCT's ME seam: parameter names / shapes it expects
    me_compat.install_as_minkowski_engine()
    ref = _load_ref_meta_arch("ref_meta_product")
    nets = {"pc_decoder": ref["pc_processor"].PC_Processor(arch_3d="MinkUNet34C"),
            "pc_binary_head": ref["pc_processor"].PC_Binary_Processor(arch_3d="MinkUNet18A")}
    keys = {n: {k: list(v.shape) for k, v in m.state_dict().items()} for n, m in nets.items()}
    with open(os.path.join(HERE, "seam_minkunet_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    print("wrote seam_minkunet_keys.json", {n: len(k) for n, k in keys.items()})

    # (2) the reference's forward code executed on the CPU stand-in of the ME ops, closed-form parameters
    me_cpu_stub.install()
    ref = _load_ref_meta_arch("ref_meta_cpu")
    coords, feats = me_cpu_stub.seam_cloud()
    arrs = dict(coords=coords, feats=feats)
    for name, cls, arch in (("pc_decoder", "PC_Processor", "MinkUNet34C"), ("pc_binary_head", "PC_Binary_Processor", "MinkUNet18A")):
        net = getattr(ref["pc_processor"], cls)(arch_3d=arch).eval()
        assert {k: list(v.shape) for k, v in net.state_dict().items()} == keys[name]
        net.load_state_dict(me_cpu_stub.closed_form_state(keys[name]))
        with torch.no_grad():
            out = net(me_cpu_stub.SparseTensor(torch.from_numpy(feats), coords))
        if name == "pc_decoder":
            arrs.update(implicit_x=out[0].numpy(), x_rows=out[1][::8].numpy(), idx=out[2].numpy())
        else:
            arrs.update(binary=out.numpy())
    save("seam_minkunet_forward.npz", **arrs)

    # (3) the MSDeformAttn extension seam: what the reference's Function passes, positionally, and what it expects back
    sys.path.insert(0, os.path.join(REF, "third_party/Mask2Former/mask2former/modeling/pixel_decoder"))
    calls = []

    class Recorder:
        """stands where the compiled extension would be; answers with the reference's own CPU formulation"""

        @staticmethod
        def ms_deform_attn_forward(*args):
            calls.append(("ms_deform_attn_forward", args))
            return core(args[0], args[1], args[3], args[4])

        @staticmethod
        def ms_deform_attn_backward(*args):
            calls.append(("ms_deform_attn_backward", args))
            v, l, w = (args[i].detach().clone().requires_grad_(True) for i in (0, 3, 4))
            with torch.enable_grad():
                core(v, args[1], l, w).backward(args[5])
            return v.grad, l.grad, w.grad

    sys.modules["MultiScaleDeformableAttention"] = Recorder
    for m in [k for k in sys.modules if k == "ops" or k.startswith("ops.")]:
        del sys.modules[m]
    from ops.functions.ms_deform_attn_func import MSDeformAttnFunction, ms_deform_attn_core_pytorch as core
    from ops.modules.ms_deform_attn import MSDeformAttn as RefModule

    torch.manual_seed(21)
    shapes = torch.as_tensor([(5, 7), (3, 4)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    value = torch.rand(2, S, 4, 8, requires_grad=True)
    loc = (torch.rand(2, 9, 4, 2, 3, 2) * 1.2 - 0.1).requires_grad_(True)
    w = torch.rand(2, 9, 4, 2, 3)
    w = (w / w.sum((-1, -2), keepdim=True)).requires_grad_(True)
    out = MSDeformAttnFunction.apply(value, shapes, lsi, loc, w, 2)
    go = torch.randn_like(out)
    out.backward(go)
    sig = {name: [("tensor", str(a.dtype).replace("torch.", ""), list(a.shape)) if torch.is_tensor(a) else ("int", int(a)) for a in args]
           for name, args in calls}
    assert set(sig) == {"ms_deform_attn_forward", "ms_deform_attn_backward"}
    refmod = RefModule(d_model=256, n_levels=3, n_heads=8, n_points=4)
    with open(os.path.join(HERE, "seam_msda_call.json"), "w") as f:
        json.dump({"calls": sig, "module_state": {k: list(v.shape) for k, v in refmod.state_dict().items()},
                   "module_im2col_step": refmod.im2col_step}, f, indent=0, sort_keys=True)
    save("seam_msda_call.npz", value=value.detach().numpy(), shapes=shapes.numpy(), level_start=lsi.numpy(), loc=loc.detach().numpy(),
         w=w.detach().numpy(), im2col_step=np.int64(2), out=out.detach().numpy(), grad_out=go.numpy(),
         g_value=value.grad.numpy(), g_loc=loc.grad.numpy(), g_w=w.grad.numpy())
    # and the product module drops into the same import name
    assert msda.install_as_msda() is sys.modules["MultiScaleDeformableAttention"]


if __name__ == "__main__":
    which = sys.argv[1:] or ["voxel", "msda", "fuser", "mapping", "diffusion", "config", "seams"]
    for w in which:
        globals()["gen_" + w]()
