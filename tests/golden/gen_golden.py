#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING THE REFERENCE on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/gen_golden.py

It imports `/root/reference/src` (read-only; `torchvision` is stubbed because
`kbnet_model.py:17` imports it for TensorBoard image grids only), feeds seeded
synthetic inputs and Philox-keyed weights through the reference's own classes
and stores {inputs, weights, outputs} as small `.npz` files next to this script.
Nothing of the reference's source is stored: the fixtures are data.

Cases
  s2d_*      networks.SparseToDensePool: pyramid (pre-hook on pool_convs) + output,
             incl. a hand-made map: isolated points, border points, an all-zero
             window, a depth of exactly 999.0 and one above 999.
  coords_*   encoder closures camera_coordinates/scale_intrinsics captured through
             forward pre-hooks on the four KB blocks (pins Q1/Q2/Q9).
  kb_*       net_utils.CalibratedBackprojectionBlock with / without `fused`, odd size; kb_stacked*: n_convolution_image /
             n_convolution_depth > 1 (stride-1 convs stacked in front of the stride-2 conv of a branch).
  dec_*      networks.MultiScaleDecoder (n_resolution=1, 'up', linear output).
  fwd_*      KBNetModel.forward: KITTI preset, VOID preset (both narrow channels)
             and an odd 70x100 frame; fwd_kb012 / fwd_kb02: encoders with KB layers at levels
             [0, 1, 2] / [0, 2] only (plain stride-2 blocks elsewhere); fwd_kb01234*: a KB layer at resolution 4 as well --
             the reference then calls calibrated_backprojection4 twice (quirk Q3; levels 2 and 3 of equal width).
  ckpt_kitti_narrow.pth  written by the reference's KBNetModel.save_model (same weights as fwd_kitti): the
             `.pth` layout restore_model must read (module.-prefixed keys, three state_dicts, optimizer state).
  io/*       input pipeline: small PNG / .npy files written by this script's own PNG writer
             (every scanline filter type, split IDAT, gray / RGB / RGBA / palette / 16-bit gray) and what the
             reference's data_utils.load_image, datasets.load_image_triplet / load_depth and
             datasets.KBNetInferenceDataset.__getitem__ read from them (io_expected.npz).
"""

import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))

import kbnet_amd as kb  # noqa: E402
import kbnet_model  # noqa: E402  (reference)
import net_utils  # noqa: E402  (reference)
import networks  # noqa: E402  (reference)

torch.set_grad_enabled(False)


def np_sd(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def load(module, sd):
    own = module.state_dict()
    pref = "module." if next(iter(own)).startswith("module.") else ""
    module.load_state_dict({pref + k: v for k, v in sd.items()}, strict=True)


def save(name, **arrays):
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}::{kk}"] = np.asarray(vv)
        else:
            flat[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def build_reference_model(cfg):
    return kbnet_model.KBNetModel(
        input_channels_image=cfg.input_channels_image,
        input_channels_depth=cfg.input_channels_depth,
        min_pool_sizes_sparse_to_dense_pool=list(cfg.min_pool_sizes_sparse_to_dense_pool),
        max_pool_sizes_sparse_to_dense_pool=list(cfg.max_pool_sizes_sparse_to_dense_pool),
        n_convolution_sparse_to_dense_pool=cfg.n_convolution_sparse_to_dense_pool,
        n_filter_sparse_to_dense_pool=cfg.n_filter_sparse_to_dense_pool,
        n_filters_encoder_image=list(cfg.n_filters_encoder_image),
        n_filters_encoder_depth=list(cfg.n_filters_encoder_depth),
        resolutions_backprojection=list(cfg.resolutions_backprojection),
        n_filters_decoder=list(cfg.n_filters_decoder),
        deconv_type=cfg.deconv_type,
        weight_initializer=cfg.weight_initializer,
        activation_func=cfg.activation_func,
        min_predict_depth=cfg.min_predict_depth,
        max_predict_depth=cfg.max_predict_depth,
        device=torch.device("cpu"))


# ------------------------------------------------------------------------- S2D
def handmade_depth():
    z = np.zeros((1, 1, 24, 30), dtype=np.float32)
    z[0, 0, 0, 0] = 3.5          # corner
    z[0, 0, 0, 17] = 12.25       # top border
    z[0, 0, 23, 29] = 7.0        # opposite corner
    z[0, 0, 11, 0] = 1.0         # left border
    z[0, 0, 10, 12] = 999.0      # exactly the sentinel value (Q6)
    z[0, 0, 10, 14] = 1500.0     # above the sentinel
    z[0, 0, 5, 6] = 2.0
    z[0, 0, 5, 7] = 40.0         # neighbours: min != max
    z[0, 0, 18, 20] = 0.00390625  # 1/256, the smallest PNG depth
    return torch.from_numpy(z)


def gen_s2d():
    for preset in ("kitti", "void"):
        cfg = kb.PRESETS[preset]()
        sd = kb.synthetic.make_state_dicts(cfg, seed=3, gain=2.0)[0]
        m = networks.SparseToDensePool(
            input_channels=cfg.input_channels_depth,
            min_pool_sizes=list(cfg.min_pool_sizes_sparse_to_dense_pool),
            max_pool_sizes=list(cfg.max_pool_sizes_sparse_to_dense_pool),
            n_convolution=cfg.n_convolution_sparse_to_dense_pool,
            n_filter=cfg.n_filter_sparse_to_dense_pool,
            weight_initializer=cfg.weight_initializer,
            activation_func=cfg.activation_func).eval()
        load(m, sd)
        captured = {}
        m.pool_convs.register_forward_pre_hook(lambda mod, args: captured.__setitem__("p", args[0].clone()))
        zs = [handmade_depth()]
        _, sp, _, _ = kb.synthetic.make_frames(2, 40, 56, preset, seed=11)
        zs.append(sp)
        # denser map so that min != max almost everywhere
        zs.append(kb.synthetic.make_frames(1, 33, 47, "kitti", seed=12)[1] *
                  (torch.rand(1, 1, 33, 47, generator=torch.Generator().manual_seed(5)) < 0.7))
        for i, z in enumerate(zs):
            x = torch.cat([z, (z > 0).float()], dim=1)
            out = m(x)
            save(f"s2d_{preset}_{i}", x=x, weights=np_sd(sd), pyramid=captured["p"], out=out,
                 min_pool_sizes=np.array(cfg.min_pool_sizes_sparse_to_dense_pool),
                 max_pool_sizes=np.array(cfg.max_pool_sizes_sparse_to_dense_pool))


# ------------------------------------------------------------------ coordinates
def gen_coords():
    cfg = kb.kitti_config().narrow()
    model = build_reference_model(cfg)
    enc = model.encoder.module if hasattr(model.encoder, "module") else model.encoder
    cases = {
        "kitti": (kb.synthetic.make_frames(2, 64, 96, "kitti", seed=21, jitter_intrinsics=0.1), 64, 96),
        "nyu": (kb.synthetic.make_frames(1, 96, 128, "nyu_v2", seed=22), 96, 128),
        "odd": (kb.synthetic.make_frames(1, 70, 100, "void", seed=23), 70, 100),
    }
    for name, ((image, sparse, valid, k), h, w) in cases.items():
        if name == "odd":
            k = k.clone()
            k[:, 0, 1] = 0.37  # skew: Q2 leaves it unscaled
        cap = {}
        hooks = []
        for lvl in range(4):
            blk = getattr(enc, f"calibrated_backprojection{lvl + 1}")
            hooks.append(blk.register_forward_pre_hook(
                (lambda lv: lambda mod, args, kwargs: cap.__setitem__(lv, kwargs["coordinates"].clone()))(lvl),
                with_kwargs=True))
        model.forward(image, sparse, valid, k)
        for hk in hooks:
            hk.remove()
        save(f"coords_{name}", intrinsics=k, height=np.array(h), width=np.array(w),
             **{f"coordinates{lv}": cap[lv] for lv in range(4)})


# --------------------------------------------------------------------- KB block
def gen_kb():
    g = torch.Generator().manual_seed(31)
    act = net_utils.activation_func("leaky_relu")
    for name, (ci, cd, cf_prev, fi, fd, ff, h, w, with_fused) in {
        "nofused": (8, 4, 0, 8, 4, 8, 20, 28, False),
        "fused": (8, 4, 8, 16, 8, 16, 18, 24, True),
        "odd": (6, 5, 7, 10, 6, 9, 13, 19, True),
    }.items():
        in_fused = ci + cf_prev if with_fused else ci
        blk = net_utils.CalibratedBackprojectionBlock(
            in_channels_image=ci, in_channels_depth=cd, in_channels_fused=in_fused,
            n_filter_image=fi, n_filter_depth=fd, n_filter_fused=ff,
            weight_initializer="xavier_normal", activation_func=act).eval()
        sd = {k: torch.randn(v.shape, generator=g) * (2.0 / (v.shape[1] * v.shape[2] * v.shape[3])) ** 0.5
              for k, v in blk.state_dict().items()}
        blk.load_state_dict(sd)
        n = 2
        image = torch.randn(n, ci, h, w, generator=g)
        depth = torch.randn(n, cd, h, w, generator=g)
        fused = torch.randn(n, cf_prev, h, w, generator=g) if with_fused else None
        k = kb.synthetic.make_frames(n, h, w, "void", seed=32, jitter_intrinsics=0.2)[3]
        k[:, 0, 0] = 30.0
        k[:, 1, 1] = 28.0
        k[:, 0, 2] = w / 2.0
        k[:, 1, 2] = h / 2.0
        xy = net_utils.meshgrid(n, h, w, device=torch.device("cpu"), homogeneous=True).view(n, 3, -1)
        coords = torch.matmul(torch.inverse(k), xy).view(n, 3, h, w)
        ci_o, cd_o, cf_o = blk(image=image, depth=depth, coordinates=coords, fused=fused)
        arrays = dict(image=image, depth=depth, coordinates=coords, intrinsics=k,
                      weights=np_sd(sd), conv_image=ci_o, conv_depth=cd_o, conv_fused=cf_o)
        if with_fused:
            arrays["fused"] = fused
        save(f"kb_{name}", **arrays)


def gen_kb_stacked():
    """n_convolution_image / n_convolution_depth > 1 (reference src/net_utils.py:1311-1325, VGGNetBlock :900-958): stride-1 3x3
    convs stacked in front of the stride-2 conv of the image / depth branch.  KBNet's presets use 1; the block takes any."""
    g = torch.Generator().manual_seed(37)
    act = net_utils.activation_func("leaky_relu")
    for name, (ci, cd, cf_prev, fi, fd, ff, h, w, n_img, n_dep) in {
        "stacked": (8, 4, 8, 16, 8, 16, 18, 24, 2, 3),
        "stacked_odd": (6, 5, 0, 10, 6, 9, 13, 19, 3, 2),
    }.items():
        with_fused = cf_prev > 0
        in_fused = ci + cf_prev if with_fused else ci
        blk = net_utils.CalibratedBackprojectionBlock(
            in_channels_image=ci, in_channels_depth=cd, in_channels_fused=in_fused,
            n_filter_image=fi, n_filter_depth=fd, n_filter_fused=ff,
            n_convolution_image=n_img, n_convolution_depth=n_dep, n_convolution_fused=2,
            weight_initializer="xavier_normal", activation_func=act).eval()
        sd = {k: torch.randn(v.shape, generator=g) * (2.0 / (v.shape[1] * v.shape[2] * v.shape[3])) ** 0.5
              for k, v in blk.state_dict().items()}
        blk.load_state_dict(sd)
        n = 2
        image = torch.randn(n, ci, h, w, generator=g)
        depth = torch.randn(n, cd, h, w, generator=g)
        fused = torch.randn(n, cf_prev, h, w, generator=g) if with_fused else None
        k = kb.synthetic.make_frames(n, h, w, "void", seed=33, jitter_intrinsics=0.2)[3]
        k[:, 0, 0] = 30.0
        k[:, 1, 1] = 28.0
        k[:, 0, 2] = w / 2.0
        k[:, 1, 2] = h / 2.0
        xy = net_utils.meshgrid(n, h, w, device=torch.device("cpu"), homogeneous=True).view(n, 3, -1)
        coords = torch.matmul(torch.inverse(k), xy).view(n, 3, h, w)
        ci_o, cd_o, cf_o = blk(image=image, depth=depth, coordinates=coords, fused=fused)
        arrays = dict(image=image, depth=depth, coordinates=coords, intrinsics=k,
                      weights=np_sd(sd), conv_image=ci_o, conv_depth=cd_o, conv_fused=cf_o,
                      n_convolution_image=np.array(n_img), n_convolution_depth=np.array(n_dep))
        if with_fused:
            arrays["fused"] = fused
        save(f"kb_{name}", **arrays)


def gen_encoder_stacked():
    """networks.KBNetEncoder with n_convolutions_* > 1 (reference src/networks.py:52-299; KBNetModel always passes ones, so this is the
    encoder's own constructor surface): stacked stride-1 convs in a KB level (level 2) and in plain VGG levels (1, 3, 4)."""
    g = torch.Generator().manual_seed(43)
    fi, fd = [8, 16, 32, 32, 32], [4, 8, 16, 16, 16]
    n_img, n_dep = [1, 2, 2, 1, 2], [1, 1, 3, 2, 1]
    enc = networks.KBNetEncoder(input_channels_image=3, input_channels_depth=8, n_filters_image=fi, n_filters_depth=fd,
                                n_filters_fused=fi, n_convolutions_image=n_img, n_convolutions_depth=n_dep,
                                n_convolutions_fused=[1, 1, 1, 1, 1], resolutions_backprojection=[0, 2],
                                weight_initializer="xavier_normal", activation_func="leaky_relu").eval()
    sd = {k: torch.randn(v.shape, generator=g) * 1.2 * (2.0 / (v.shape[1] * v.shape[2] * v.shape[3])) ** 0.5
          for k, v in enc.state_dict().items()}
    enc.load_state_dict(sd)
    n, h, w = 2, 38, 52
    image = torch.rand(n, 3, h, w, generator=g)
    depth = torch.nn.functional.leaky_relu(torch.randn(n, 8, h, w, generator=g), 0.2)
    k = kb.synthetic.make_frames(n, h, w, "void", seed=44, jitter_intrinsics=0.1)[3]
    k[:, 0, 0] = 60.0; k[:, 1, 1] = 58.0; k[:, 0, 2] = w / 2.0; k[:, 1, 2] = h / 2.0
    latent, skips = enc(image, depth, k)
    save("enc_stacked", image=image, depth=depth, intrinsics=k, weights=np_sd(sd), latent=latent,
         n_convolutions_image=np.array(n_img), n_convolutions_depth=np.array(n_dep), n_filters_image=np.array(fi), n_filters_depth=np.array(fd),
         resolutions_backprojection=np.array([0, 2]), **{f"skip{i + 1}": s_ for i, s_ in enumerate(skips)})


# ---------------------------------------------------------------------- decoder
def gen_decoder(only=None):
    import dataclasses
    # dec_transpose: deconv_type='transpose' (run_kbnet.py --deconv_type transpose): TransposeConv2d in every block
    # (src/net_utils.py:350-440); even sizes only -- the transposed conv doubles the size whatever the skip's is
    for deconv_type, cases in (("up", {"even": (64, 96), "odd": (70, 100)}), ("transpose", {"transpose": (64, 96)})):
        cfg = dataclasses.replace(kb.kitti_config().narrow(), deconv_type=deconv_type)
        sd = kb.synthetic.make_state_dicts(cfg, seed=4, gain=1.5)[2]
        enc_ch = [i + z for i, z in zip(cfg.n_filters_encoder_image, cfg.n_filters_encoder_depth)]
        dec = networks.MultiScaleDecoder(
            input_channels=enc_ch[-1], output_channels=1, n_resolution=1,
            n_filters=list(cfg.n_filters_decoder), n_skips=cfg.n_skips,
            weight_initializer="xavier_normal", activation_func="leaky_relu",
            output_func="linear", use_batch_norm=False, deconv_type=deconv_type).eval()
        load(dec, sd)
        g = torch.Generator().manual_seed(41)
        for name, (h, w) in cases.items():
            sizes = [(h, w)]
            for _ in range(5):
                sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
            latent = torch.randn(1, enc_ch[4], *sizes[5], generator=g)
            skips = [torch.randn(1, enc_ch[i], *sizes[i + 1], generator=g) for i in range(4)]
            if only and name not in only:
                continue
            out = dec(latent, skips, (h, w))[-1]
            save(f"dec_{name}", latent=latent, weights=np_sd(sd), logits=out, shape=np.array([h, w]),
                 **{f"skip{i + 1}": s for i, s in enumerate(skips)})


# ----------------------------------------------------------------- full forward
def gen_forward(only=None):
    import dataclasses
    for name, preset, (h, w), n, kb_levels in (("kitti", "kitti", (64, 96), 2, None),
                                               ("void", "void", (96, 128), 1, None),
                                               ("odd", "void", (70, 100), 1, None),
                                               # encoder topologies other than KBNet's [0, 1, 2, 3]: plain stride-2
                                               # VGG blocks where a level has no KB layer (src/networks.py:150-299)
                                               ("kb012", "kitti", (64, 96), 2, (0, 1, 2)),
                                               ("kb02", "void", (48, 80), 1, (0, 2)),
                                               # KB layer at resolution 4 too: the reference's level-4 branch calls
                                               # calibrated_backprojection4 a second time (src/networks.py:499-517, quirk Q3),
                                               # which only runs when levels 2 and 3 have the same widths
                                               ("kb01234", "kitti", (64, 96), 2, (0, 1, 2, 3, 4)),
                                               ("kb01234_odd", "void", (70, 100), 1, (0, 1, 2, 3, 4)),
                                               # deconv_type='transpose' (run_kbnet.py --deconv_type): the decoder's up-sampling
                                               # layers are ConvTranspose2d(3, stride 2, padding 1, output_padding 1)
                                               ("transpose", "kitti", (64, 96), 2, None),
                                               ("transpose_void", "void", (96, 128), 1, None),
                                               # run_kbnet.py --activation_func: every factory branch of net_utils.activation_func
                                               # (src/net_utils.py:23-45) other than the default leaky_relu
                                               ("act_relu", "kitti", (64, 96), 1, None),
                                               ("act_elu", "kitti", (64, 96), 1, None),
                                               ("act_sigmoid", "void", (70, 100), 1, None),
                                               ("act_linear", "kitti", (64, 96), 1, None),
                                               ("act_elu_kb012_transpose", "void", (64, 96), 1, (0, 1, 2))):
        if only and name not in only:
            continue
        cfg = kb.PRESETS[preset]().narrow()
        if "transpose" in name:
            cfg = dataclasses.replace(cfg, deconv_type="transpose")
        if name.startswith("act_"):
            cfg = dataclasses.replace(cfg, activation_func=name.split("_")[1])
        if kb_levels is not None:
            cfg = dataclasses.replace(cfg, resolutions_backprojection=kb_levels)
        if kb_levels is not None and 4 in kb_levels:
            cfg = dataclasses.replace(cfg, n_filters_encoder_image=(8, 16, 32, 32, 32), n_filters_encoder_depth=(4, 8, 16, 16, 16))
        # gain > 1 keeps the logits O(1) (random xavier weights shrink the signal), so the
        # sigmoid head is exercised off its saturated ends
        gain = 1.3 if preset == "kitti" else 1.45
        if name.startswith("act_"):   # the same off-saturation, O(1) logits for the other activations (a linear net grows like gain^21)
            gain = {"linear": 0.75, "relu": 1.2, "elu": 1.2, "sigmoid": 1.45}[cfg.activation_func]
        sds = kb.synthetic.make_state_dicts(cfg, seed=5, gain=gain)
        model = build_reference_model(cfg)
        model.eval()
        load(model.sparse_to_dense_pool, sds[0])
        load(model.encoder, sds[1])
        load(model.decoder, sds[2])
        density_kind = preset
        image, sparse, valid, k = kb.synthetic.make_frames(n, h, w, density_kind, seed=51,
                                                           jitter_intrinsics=0.05)
        if preset == "void":
            # denser than VOID so small frames still hold a few dozen points
            g = np.random.Generator(np.random.Philox(52))
            m = torch.from_numpy((g.random((n, 1, h, w), dtype=np.float32) < 0.03))
            d = torch.from_numpy(np.round((0.3 + 4.7 * g.random((n, 1, h, w), dtype=np.float32)) * 256) / 256)
            sparse = (d * m).float()
            valid = (sparse > 0).float()
        out = model.forward(image, sparse, valid, k)
        save(f"fwd_{name}", image=image, sparse_depth=sparse, validity_map=valid, intrinsics=k,
             output_depth=out, preset=np.array(preset), deconv_type=np.array(cfg.deconv_type), activation_func=np.array(cfg.activation_func),
             resolutions_backprojection=np.array(cfg.resolutions_backprojection),
             n_filters_encoder_image=np.array(cfg.n_filters_encoder_image), n_filters_encoder_depth=np.array(cfg.n_filters_encoder_depth),
             s2d=np_sd(sds[0]), encoder=np_sd(sds[1]), decoder=np_sd(sds[2]))


# ------------------------------------------------------- checkpoint layout (f3)
def gen_checkpoint():
    """A checkpoint written by the REFERENCE's own writer (KBNetModel.save_model, src/kbnet_model.py:353-376):
    the narrow KITTI-preset model of fwd_kitti (same weights), an untouched Adam optimizer as the training loop
    passes one (src/kbnet.py:507-518), train_step 1234.  The file is what restore_model (:378-406) reads."""
    cfg = kb.PRESETS["kitti"]().narrow()
    sds = kb.synthetic.make_state_dicts(cfg, seed=5, gain=1.3)
    model = build_reference_model(cfg)
    load(model.sparse_to_dense_pool, sds[0])
    load(model.encoder, sds[1])
    load(model.decoder, sds[2])
    optimizer = torch.optim.Adam([{"params": model.parameters(), "weight_decay": 0.0}], lr=1e-4)
    path = os.path.join(HERE, "ckpt_kitti_narrow.pth")
    model.save_model(path, 1234, optimizer)
    ckpt = torch.load(path)
    assert set(ckpt) == {"train_step", "optimizer_state_dict", "sparse_to_dense_pool_state_dict",
                         "encoder_state_dict", "decoder_state_dict"}
    assert all(k.startswith("module.") for k in ckpt["encoder_state_dict"])
    print(f"ckpt_kitti_narrow.pth: {os.path.getsize(path) / 1024:.1f} KiB, "
          f"{sum(v.numel() for d in list(ckpt.values())[2:] for v in d.values())} parameters")


# ------------------------------------------- pre-model stage and evaluation (f1, f2)
def gen_pre_eval():
    import eval_utils  # reference
    g = np.random.Generator(np.random.Philox(71))
    n, h, w = 2, 40, 64
    sparse = kb.synthetic.make_frames(n, h, w, "kitti", seed=72)[1]
    dense_mask = torch.from_numpy(g.random((n, 1, h, w), dtype=np.float32) < 0.25)
    sparse = torch.where(dense_mask, torch.from_numpy(np.round((5 + 3 * g.random((n, 1, h, w), dtype=np.float32)) * 256) / 256), sparse)
    # outliers: far points sitting inside a near neighbourhood
    sparse[0, 0, 10, 10] = 60.0
    sparse[1, 0, 0, 0] = 75.5
    sparse[1, 0, 39, 63] = 0.5
    image = torch.from_numpy(np.floor(256 * g.random((n, 3, h, w), dtype=np.float32)))
    validity = torch.where(sparse > 0, torch.ones_like(sparse), sparse)           # src/kbnet.py:899-902
    removal = net_utils.OutlierRemoval(7, 1.5)
    f_sparse, f_valid = removal.remove_outliers(sparse_depth=sparse, validity_map=validity)
    import transforms as ref_transforms  # reference: Transforms.normalize_images, src/transforms.py:185-214
    tr = ref_transforms.Transforms(normalized_image_range=[0, 1])
    image_n = tr.normalize_images([image], [0, 1])[0]
    assert torch.equal(image_n, image / 255.0)
    image_m = tr.normalize_images([image], [-1, 1])[0]                           # run_kbnet.py --normalized_image_range -1 1
    save("pre_outlier", image=image, sparse_depth=sparse, validity_map=validity, filtered_sparse_depth=f_sparse,
         filtered_validity_map=f_valid, image_normalized=image_n, image_normalized_m1_1=image_m,
         kernel_size=np.array(7), threshold=np.array(1.5))

    # evaluation: reference src/kbnet.py:932-950 on one frame
    h, w = 48, 80
    gt = (1.0 + 79.0 * g.random((h, w), dtype=np.float32)).astype(np.float32)
    gtv = (g.random((h, w), dtype=np.float32) < 0.3).astype(np.float32)
    gt = gt * gtv
    pred = (gt + g.standard_normal((h, w)).astype(np.float32) * 0.8 + 2.0 * (1 - gtv)).clip(1.5, 100).astype(np.float32)
    vmask = np.where(gtv > 0, 1, 0)
    mm = np.logical_and(gt > 0.0, gt < 100.0)
    mask = np.where(np.logical_and(vmask, mm) > 0)
    o, t = pred[mask], gt[mask]
    metrics = np.array([eval_utils.mean_abs_err(1000.0 * o, 1000.0 * t), eval_utils.root_mean_sq_err(1000.0 * o, 1000.0 * t),
                        eval_utils.inv_mean_abs_err(0.001 * o, 0.001 * t), eval_utils.inv_root_mean_sq_err(0.001 * o, 0.001 * t)],
                       dtype=np.float64)
    save("eval_metrics", output_depth=pred, ground_truth=gt, validity_map=gtv, metrics=metrics,
         min_evaluate_depth=np.array(0.0), max_evaluate_depth=np.array(100.0))


def _png_bytes(arr, color_type, palette=None, filters=(0, 1, 2, 3, 4), idat_split=1):
    """Minimal PNG writer: `arr` H x W (x C) uint8 or H x W uint16; the scanline filter cycles through
    `filters` row by row so that every filter type of the spec appears in the fixtures."""
    import struct
    import zlib
    h, w = arr.shape[:2]
    bit_depth = 16 if arr.dtype == np.uint16 else 8
    rows = arr.astype(">u2").tobytes() if bit_depth == 16 else arr.astype(np.uint8).tobytes()
    bpp = (arr.shape[2] if arr.ndim == 3 else 1) * (bit_depth // 8)
    stride = w * bpp
    raw, prev = bytearray(), bytes(stride)
    for y in range(h):
        cur = rows[y * stride:(y + 1) * stride]
        f = filters[y % len(filters)]
        out = bytearray(stride)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 0:
                pred = 0
            elif f == 1:
                pred = a
            elif f == 2:
                pred = b
            elif f == 3:
                pred = (a + b) >> 1
            else:
                p_ = a + b - c
                pa, pb, pc = abs(p_ - a), abs(p_ - b), abs(p_ - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (cur[i] - pred) & 255
        raw += bytes([f]) + out
        prev = cur

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    z = zlib.compress(bytes(raw), 6)
    cuts = [len(z) * i // idat_split for i in range(idat_split + 1)]
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, color_type, 0, 0, 0))
    if palette is not None:
        png += chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    for i in range(idat_split):
        png += chunk(b"IDAT", z[cuts[i]:cuts[i + 1]])
    return png + chunk(b"IEND", b"")


def gen_io():
    import data_utils  # noqa: E402  (reference)
    import datasets  # noqa: E402  (reference)
    d = os.path.join(HERE, "io")
    os.makedirs(d, exist_ok=True)
    g = np.random.Generator(np.random.Philox(77))
    h, w = 24, 40
    exp = {}

    def put(name, data):
        with open(os.path.join(d, name), "wb") as f:
            f.write(data)
        return os.path.join(d, name)

    image_paths, depth_paths, k_paths = [], [], []
    for i in range(3):
        trip = g.integers(0, 256, size=(h, 3 * w, 3), dtype=np.uint8)
        image_paths.append(put(f"triplet_{i}.png", _png_bytes(trip, 2, filters=(i, 1, 2, 3, 4, 0), idat_split=1 + i)))
        dep = (g.integers(0, 65536, size=(h, w)) * (g.random((h, w)) < 0.2)).astype(np.uint16)
        dep[0, 0], dep[0, 1], dep[1, 0] = 65535, 1, 256
        depth_paths.append(put(f"depth_{i}.png", _png_bytes(dep, 0, filters=(4, 3, 2, 1, 0), idat_split=2)))
        k = np.array([[721.5377 + i, 0, 609.5593], [0, 721.5377, 172.854 - i], [0, 0, 1]], dtype=np.float64)
        np.save(os.path.join(d, f"k_{i}.npy"), k)
        k_paths.append(os.path.join(d, f"k_{i}.npy"))
    gray = put("gray8.png", _png_bytes(g.integers(0, 256, size=(10, 12), dtype=np.uint8), 0))
    rgba = put("rgba.png", _png_bytes(g.integers(0, 256, size=(7, 9, 4), dtype=np.uint8), 6, filters=(4, 4, 1)))
    pal = put("palette.png", _png_bytes(g.integers(0, 16, size=(6, 8), dtype=np.uint8), 3,
                                        palette=g.integers(0, 256, size=(16, 3), dtype=np.uint8), filters=(0, 2)))
    depth8 = put("depth8.png", _png_bytes(g.integers(0, 256, size=(5, 8), dtype=np.uint8), 0, filters=(1,)))

    # what the reference reads from those files
    for name, path in (("gray8", gray), ("rgba", rgba), ("palette", pal)):
        exp[f"load_image_{name}_hwc"] = data_utils.load_image(path, normalize=True, data_format="HWC")
        exp[f"load_image_{name}_chw_raw"] = data_utils.load_image(path, normalize=False, data_format="CHW")
    t1, t0, t2 = datasets.load_image_triplet(image_paths[0], normalize=True)
    exp["triplet_0_t"], exp["triplet_0_tm1"], exp["triplet_0_tp1"] = t0, t1, t2
    exp["load_depth_0_hw"] = data_utils.load_depth(depth_paths[0], data_format="HW")
    exp["load_depth8_chw"] = data_utils.load_depth(depth8, data_format="CHW")
    z, v = data_utils.load_depth_with_validity_map(depth_paths[1], data_format="CHW")
    exp["depth_1_z"], exp["depth_1_v"] = z, v
    ds = datasets.KBNetInferenceDataset(image_paths=image_paths, sparse_depth_paths=depth_paths,
                                        intrinsics_paths=k_paths, use_image_triplet=True)
    for i in range(len(ds)):
        image, sparse_depth, intrinsics = ds[i]
        exp[f"sample_{i}_image"], exp[f"sample_{i}_sparse_depth"], exp[f"sample_{i}_intrinsics"] = image, sparse_depth, intrinsics
    ds1 = datasets.KBNetInferenceDataset(image_paths=[rgba], sparse_depth_paths=[depth8], intrinsics_paths=k_paths[:1],
                                         use_image_triplet=False)
    exp["single_image"] = ds1[0][0]
    np.savez_compressed(os.path.join(HERE, "io_expected.npz"), **exp)
    print("io: %d files, %d expected arrays" % (len(os.listdir(d)), len(exp)))


def gen_saved_depth():
    """The output side: a depth map written by the REFERENCE's data_utils.save_depth (src/data_utils.py:154-167), the file and what
    the reference's load_depth reads back from it."""
    import data_utils  # noqa: E402  (reference)
    g = np.random.Generator(np.random.Philox(91))
    h, w = 37, 50
    z = (g.random((h, w), dtype=np.float32) * 120.0 * (g.random((h, w)) < 0.7)).astype(np.float32)
    z[0, :6] = [0.0, 1.5, 100.0, 255.99, 256.0, 300.0]          # the last two lie past the 16 bits of the PNG: PIL clips them
    z[1, :3] = [1.0 / 256.0, 0.999 / 256.0, 65535.0 / 256.0]
    path = os.path.join(HERE, "io", "saved_depth.png")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        data_utils.save_depth(z, path)
    back = data_utils.load_depth(path, data_format="HW")
    np.savez_compressed(os.path.join(HERE, "saved_depth.npz"), z=z, loaded=back)
    print("saved_depth: %d bytes, max sample %d" % (os.path.getsize(path), int(back.max() * 256)))


if __name__ == "__main__":
    if "--only-save-depth" in sys.argv:   # round 5: the writer of run_kbnet.py --save_outputs
        gen_saved_depth()
        sys.exit(0)
    if "--only-io" in sys.argv:
        gen_io()
        sys.exit(0)
    if "--only-checkpoint" in sys.argv:   # added in round 2; leaves the other fixtures untouched
        gen_checkpoint()
        sys.exit(0)
    if "--only-topologies" in sys.argv:   # added later; leaves the other fixtures untouched
        gen_forward(only=("kb012", "kb02"))
        sys.exit(0)
    if "--only-round5" in sys.argv:       # round 5: KB layer at resolution 4 (quirk Q3), stacked convs in the KB block
        gen_forward(only=("kb01234", "kb01234_odd"))
        gen_kb_stacked()
        gen_encoder_stacked()
        sys.exit(0)
    if "--only-transpose" in sys.argv:    # round 5: deconv_type='transpose'
        gen_decoder(only=("transpose",))
        gen_forward(only=("transpose", "transpose_void"))
        sys.exit(0)
    if "--only-activations" in sys.argv:  # round 5: activation_func other than leaky_relu
        gen_forward(only=("act_relu", "act_elu", "act_sigmoid", "act_linear", "act_elu_kb012_transpose"))
        sys.exit(0)
    gen_pre_eval()
    if "--only-pre-eval" in sys.argv:
        sys.exit(0)
    gen_s2d()
    gen_coords()
    gen_kb()
    gen_kb_stacked()
    gen_encoder_stacked()
    gen_decoder()
    gen_forward()
    gen_checkpoint()
    gen_saved_depth()
