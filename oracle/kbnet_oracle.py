"""CPU oracle for the KBNet inference hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference algorithm
`KBNetModel.forward` (reference `src/kbnet_model.py:143-186`) and everything below
it.  It exists to CHECK the HIP path; it is never the thing that is shipped or
measured as the product:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import it;
  * the product package (`calibrated-backprojection-network_amd/`) does not
    import it and raises when its HIP extension is missing.

Pinning (see DESIGN.md "Oracle"): the reference holds no tests or golden
vectors for this path.  The oracle is pinned against the
reference ITSELF: `tests/golden/gen_golden.py` imports `/root/reference/src`
in the build container, runs it on seeded inputs and commits the input/output
vectors under `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this
file against those vectors bit-for-bit, and `tests/test_oracle_vs_reference.py`
(skipped where `/root/reference` is absent) repeats the comparison at the full
352x1216 / 480x640 sizes.

Weights are passed as plain dicts keyed exactly like the reference's
`state_dict()` (without the DataParallel `module.` prefix, which `strip_prefix`
removes), all OIHW fp32, bias-free.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

NEGATIVE_SLOPE = 0.20  # reference src/net_utils.py:36-37 ('leaky_relu' factory)


def strip_prefix(state_dict, prefix="module."):
    """Checkpoint keys carry `module.` (reference src/kbnet_model.py:392-397)."""
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state_dict.items()}


def activation_slope(activation_fn):
    """The `slope` argument of every function below for the reference's activation_func STRING (src/net_utils.py:23-45, same
    substring tests in the same order): 'linear' -> None, 'leaky_relu' -> 0.20, 'relu' -> 0.0 (a LeakyReLU of slope 0 differs from
    ReLU in the sign of zeros only), 'elu' / 'sigmoid' -> that name."""
    if "linear" in activation_fn:
        return None
    if "leaky_relu" in activation_fn:
        return NEGATIVE_SLOPE
    if "relu" in activation_fn:
        return 0.0
    if "elu" in activation_fn:
        return "elu"
    if "sigmoid" in activation_fn:
        return "sigmoid"
    raise ValueError("Unsupported activation function: {}".format(activation_fn))


def activate(y, slope):
    """`slope`: None (no activation), a LeakyReLU slope, or 'elu' / 'sigmoid' (torch.nn.ELU() / torch.nn.Sigmoid(), reference
    src/net_utils.py:38-43)."""
    if slope is None:
        return y
    if slope == "elu":
        return F.elu(y)
    if slope == "sigmoid":
        return torch.sigmoid(y)
    return F.leaky_relu(y, negative_slope=slope)


def conv2d(x, weight, stride=1, slope=NEGATIVE_SLOPE):
    """Bias-free conv, padding k//2, optional activation (`slope`: see `activate`).

    Reference: net_utils.Conv2d, src/net_utils.py:85-93 (ctor) and :120-141
    (forward).  `slope=None` is the reference's activation_func=None.
    """
    k = weight.shape[-1]
    y = F.conv2d(x, weight, bias=None, stride=stride, padding=k // 2)
    return activate(y, slope)


# --------------------------------------------------------------------------- S2D
def s2d_pool_pyramid(z, min_pool_sizes, max_pool_sizes):
    """Multi-scale min/max pyramid of the sparse depth channel.

    Reference: SparseToDensePool.forward, src/networks.py:2170-2189, with the
    pool construction of :2112-2132 (sizes <= 1 dropped, stride 1, pad k//2).
    Min pool of non-zeros = -maxpool(where(z == 0, -999, -z)), 999 -> 0.
    """
    pyramid = []
    for s in [s for s in min_pool_sizes if s > 1]:
        neg = torch.where(z == 0, torch.full_like(z, -999.0), -z)
        p = -F.max_pool2d(neg, kernel_size=s, stride=1, padding=s // 2)
        p = torch.where(p == 999, torch.zeros_like(z), p)
        pyramid.append(p)
    for s in [s for s in max_pool_sizes if s > 1]:
        pyramid.append(F.max_pool2d(z, kernel_size=s, stride=1, padding=s // 2))
    return torch.cat(pyramid, dim=1)


def sparse_to_dense_pool(x, sd, min_pool_sizes, max_pool_sizes, slope=NEGATIVE_SLOPE,
                         return_pyramid=False):
    """S2D: pyramid -> n x conv1x1 -> cat[., x] -> conv3x3.

    Reference: src/networks.py:2168-2196.  `sd` keys: `pool_convs.{i}.conv.weight`,
    `conv.conv.weight`.
    """
    z = x[:, 0:1]
    pyramid = s2d_pool_pyramid(z, min_pool_sizes, max_pool_sizes)
    h = pyramid
    i = 0
    while f"pool_convs.{i}.conv.weight" in sd:
        h = conv2d(h, sd[f"pool_convs.{i}.conv.weight"], 1, slope)
        i += 1
    h = torch.cat([h, x], dim=1)
    out = conv2d(h, sd["conv.conv.weight"], 1, slope)
    return (pyramid, out) if return_pyramid else out


# ------------------------------------------------------------------- coordinates
def pixel_grid(n_batch, height, width):
    """N x 3 x H x W homogeneous pixel grid (x, y, 1), x in [0, W-1], y in [0, H-1].

    Reference: net_utils.meshgrid, src/net_utils.py:1620-1634.
    """
    xs = torch.linspace(start=0.0, end=width - 1, steps=width)
    ys = torch.linspace(start=0.0, end=height - 1, steps=height)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    grid = torch.stack([gx, gy, torch.ones_like(gx)], dim=0)
    return grid.unsqueeze(0).repeat(n_batch, 1, 1, 1)


def camera_coordinates(k, height, width):
    """K^-1 [x y 1]^T for every pixel.  Reference: src/networks.py:317-331."""
    n = k.shape[0]
    xy_h = pixel_grid(n, height, width).view(n, 3, -1)
    return torch.matmul(torch.inverse(k), xy_h).view(n, 3, height, width)


def scale_intrinsics(k, height0, width0, height1, width1):
    """K (.) [[sx,1,sx],[1,sy,sy],[1,1,1]], sx = w1/w0, sy = h1/h0.

    Reference: src/networks.py:333-352.  NOTE the caller (`encoder`) always
    passes the LEVEL-1 size as (height1, width1): the reference's closure reads
    `n_width1/n_height1` instead of its arguments (:342-343), so KB levels 2, 3
    and 4 are all scaled by the half-resolution ratio.
    """
    sx = width1 / width0
    sy = height1 / height0
    scale = torch.tensor([[sx, 1.0, sx], [1.0, sy, sy], [1.0, 1.0, 1.0]], dtype=torch.float32)
    return k * scale.view(1, 3, 3)


# -------------------------------------------------------------------- VGG block
def vgg_block(x, sd, prefix, stride, slope=NEGATIVE_SLOPE):
    """VGGNetBlock.forward, reference src/net_utils.py:900-958: n_convolution - 1 stride-1 3x3 convs, then one 3x3 conv
    with `stride` (keys `<prefix>.conv_block.<i>.conv.weight`, i = 0 .. n_convolution - 1); KBNet's presets use one conv."""
    n = 0
    while f"{prefix}.conv_block.{n}.conv.weight" in sd:
        n += 1
    if n == 0:
        raise KeyError(f"{prefix}.conv_block.0.conv.weight")
    for i in range(n):
        x = conv2d(x, sd[f"{prefix}.conv_block.{i}.conv.weight"], stride if i == n - 1 else 1, slope)
    return x


# --------------------------------------------------------------------- KB block
def kb_block(image, depth, coordinates, fused, sd, slope=NEGATIVE_SLOPE):
    """Calibrated backprojection block.

    Reference: CalibratedBackprojectionBlock.forward, src/net_utils.py:1343-1371.
    `sd` keys: conv_image.conv_block.<i>.conv.weight, conv_depth.conv_block.<i>.conv.weight (one conv each in KBNet's
    presets; n_convolution_image / n_convolution_depth > 1 stack stride-1 convs in front, :1311-1325),
    proj_depth.conv.weight, conv_fused.conv.weight.
    Returns (conv_image, conv_depth, conv_fused), all at ceil(H/2) x ceil(W/2).
    """
    conv_image = vgg_block(image, sd, "conv_image", 2, slope)
    conv_depth = vgg_block(torch.cat([depth, coordinates], dim=1), sd, "conv_depth", 2, slope)
    z = conv2d(depth, sd["proj_depth.conv.weight"], 1, slope)
    xyz = coordinates * z
    layers = [image, xyz] + ([fused] if fused is not None else [])
    conv_fused = conv2d(torch.cat(layers, dim=1), sd["conv_fused.conv.weight"], 2, slope)
    return conv_image, conv_depth, conv_fused


def _sub(sd, prefix):
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


# ---------------------------------------------------------------------- encoder
def encoder(image, depth, intrinsics, sd, resolutions_backprojection=(0, 1, 2, 3),
            slope=NEGATIVE_SLOPE, return_trace=False):
    """KBNetEncoder.forward, reference src/networks.py:301-533.

    Level 0 must be a KB level (the reference is undefined otherwise, SURVEY.md
    Q4); levels 1..3 are KB levels when listed, plain stride-2 VGG blocks
    otherwise; level 4 is plain in the shipped configs.  With 4 listed the
    reference's level-4 branch calls `calibrated_backprojection4` AGAIN
    (src/networks.py:499-517, quirk Q3: the block it built as
    `calibrated_backprojection5` is never used), which only runs when level 3 is
    a KB level too and levels 2 and 3 have the same widths -- reproduced here.
    Returns (latent, [skip1..skip4]).
    """
    if 0 not in resolutions_backprojection:
        raise ValueError("resolution 0 must use calibrated backprojection (reference Q4)")
    if 4 in resolutions_backprojection and 3 not in resolutions_backprojection:
        raise ValueError("resolution 4 re-uses calibrated_backprojection4 (reference Q3), which exists only with resolution 3")
    n, _, h0, w0 = image.shape
    trace = {}

    coords = camera_coordinates(intrinsics, h0, w0)
    conv_image = conv2d(image, sd["conv0_image.conv.weight"], 1, slope)
    conv_depth = conv2d(depth, sd["conv0_depth.conv.weight"], 1, slope)
    trace["coordinates0"] = coords
    conv_image, conv_depth, conv_fused = kb_block(
        conv_image, conv_depth, coords, None, _sub(sd, "calibrated_backprojection1"), slope)
    skips = [torch.cat([conv_fused, conv_depth], dim=1)]
    h1, w1 = conv_image.shape[-2:]

    for level in (1, 2, 3):
        hl, wl = conv_image.shape[-2:]
        if level in resolutions_backprojection:
            k_l = scale_intrinsics(intrinsics, h0, w0, h1, w1)  # Q1: always level-1 ratio
            coords = camera_coordinates(k_l, hl, wl)
            trace[f"intrinsics{level}"] = k_l
            trace[f"coordinates{level}"] = coords
            conv_image, conv_depth, conv_fused = kb_block(
                conv_image, conv_depth, coords, conv_fused,
                _sub(sd, f"calibrated_backprojection{level + 1}"), slope)
            skips.append(torch.cat([conv_fused, conv_depth], dim=1))
        else:
            src = conv_fused if conv_fused is not None else conv_image
            conv_image = vgg_block(src, sd, f"conv{level + 1}_image", 2, slope)
            conv_depth = vgg_block(conv_depth, sd, f"conv{level + 1}_depth", 2, slope)
            conv_fused = None
            skips.append(torch.cat([conv_image, conv_depth], dim=1))

    if 4 in resolutions_backprojection:
        hl, wl = conv_image.shape[-2:]
        k_l = scale_intrinsics(intrinsics, h0, w0, h1, w1)  # Q1: always level-1 ratio
        coords = camera_coordinates(k_l, hl, wl)
        trace["intrinsics4"] = k_l
        trace["coordinates4"] = coords
        _, conv5_depth, conv5_fused = kb_block(conv_image, conv_depth, coords, conv_fused,
                                               _sub(sd, "calibrated_backprojection4"), slope)   # Q3: block 4 again
        latent = torch.cat([conv5_fused, conv5_depth], dim=1)
    else:
        src = conv_fused if conv_fused is not None else conv_image
        conv5_image = vgg_block(src, sd, "conv5_image", 2, slope)
        conv5_depth = vgg_block(conv_depth, sd, "conv5_depth", 2, slope)
        latent = torch.cat([conv5_image, conv5_depth], dim=1)
    if return_trace:
        return latent, skips, trace
    return latent, skips


# ---------------------------------------------------------------------- decoder
def decoder_block(x, skip, shape, sd, slope=NEGATIVE_SLOPE):
    """Nearest-resize -> conv3x3 -> cat skip -> conv3x3 (deconv_type='up'), or ConvTranspose2d(3, stride 2, padding 1,
    output_padding 1) -> activation -> cat skip -> conv3x3 (deconv_type='transpose': the state dict then holds
    `deconv.deconv.weight`, in x out x 3 x 3).

    Reference: DecoderBlock.forward src/net_utils.py:1453-1487, UpConv2d.forward :484-499,
    TransposeConv2d :383-390 (ctor) and :417-437 (forward).
    """
    if "deconv.deconv.weight" in sd:
        y = F.conv_transpose2d(x, sd["deconv.deconv.weight"], bias=None, stride=2, padding=1, output_padding=1)
        y = activate(y, slope)
    else:
        if skip is not None:
            shape = skip.shape[2:4]
        up = F.interpolate(x, size=tuple(shape), mode="nearest")
        y = conv2d(up, sd["deconv.conv.conv.weight"], 1, slope)
    if skip is not None:
        y = torch.cat([y, skip], dim=1)
    return conv2d(y, sd["conv.conv.weight"], 1, slope)


def decoder(latent, skips, shape, sd, slope=NEGATIVE_SLOPE):
    """MultiScaleDecoder.forward with n_resolution=1, output_func='linear',
    deconv_type 'up' or 'transpose' -- whichever the state dict holds (reference
    src/networks.py:1855-1989 as configured at src/kbnet_model.py:127-137).
    Returns the full-resolution logits."""
    x = latent
    for name, skip in (("deconv4", skips[3]), ("deconv3", skips[2]),
                       ("deconv2", skips[1]), ("deconv1", skips[0])):
        x = decoder_block(x, skip, None, _sub(sd, name), slope)
    x = decoder_block(x, None, shape, _sub(sd, "deconv0"), slope)
    return conv2d(x, sd["output0.conv.weight"], 1, None)


# ---------------------------------------------------------------------- forward
def depth_head(logits, min_predict_depth, max_predict_depth):
    """sigmoid then d_min / (s + d_min/d_max).  Reference src/kbnet_model.py:181-184."""
    s = torch.sigmoid(logits)
    return min_predict_depth / (s + min_predict_depth / max_predict_depth)


def kbnet_forward(image, sparse_depth, validity_map_depth, intrinsics,
                  sd_s2d, sd_encoder, sd_decoder,
                  min_pool_sizes, max_pool_sizes,
                  min_predict_depth, max_predict_depth,
                  resolutions_backprojection=(0, 1, 2, 3), slope=NEGATIVE_SLOPE):
    """KBNetModel.forward, reference src/kbnet_model.py:143-186."""
    with torch.no_grad():
        x = torch.cat([sparse_depth, validity_map_depth], dim=1)
        d = sparse_to_dense_pool(x, strip_prefix(sd_s2d), min_pool_sizes, max_pool_sizes, slope)
        latent, skips = encoder(image, d, intrinsics, strip_prefix(sd_encoder),
                                resolutions_backprojection, slope)
        logits = decoder(latent, skips, d.shape[-2:], strip_prefix(sd_decoder), slope)
        return depth_head(logits, min_predict_depth, max_predict_depth)


# ------------------------------------------------------- "next" rows
def validity_and_outlier_removal(sparse_depth, kernel_size=7, threshold=1.5):
    """Validity map + outlier removal, reference src/kbnet.py:899-908 and
    OutlierRemoval.remove_outliers src/net_utils.py:1761-1806.  Note the
    batch-global `10 * max(sparse_depth)` fill value (:1776)."""
    validity = torch.where(sparse_depth > 0, torch.ones_like(sparse_depth), sparse_depth)
    max_value = 10 * torch.max(sparse_depth)
    filled = torch.where(validity <= 0, torch.full_like(sparse_depth, fill_value=float(max_value)),
                         sparse_depth)
    pad = kernel_size // 2
    filled = F.pad(filled, (pad, pad, pad, pad), mode="constant", value=float(max_value))
    min_values = -F.max_pool2d(-filled, kernel_size=kernel_size, stride=1, padding=0)
    keep = torch.where(min_values < sparse_depth - threshold,
                       torch.zeros_like(validity), torch.ones_like(validity))
    validity_clean = validity * keep
    return sparse_depth * validity_clean, validity_clean


def evaluation_metrics(output_depth, ground_truth, validity_map, min_evaluate_depth, max_evaluate_depth):
    """(MAE mm, RMSE mm, iMAE 1/km, iRMSE 1/km) of ONE frame, in numpy like the reference:
    mask and scaling from src/kbnet.py:932-950, formulas from src/eval_utils.py:20-78."""
    import numpy as np
    o = np.squeeze(np.asarray(output_depth))
    g = np.squeeze(np.asarray(ground_truth))
    v = np.squeeze(np.asarray(validity_map))
    validity_mask = np.where(v > 0, 1, 0)
    min_max_mask = np.logical_and(g > min_evaluate_depth, g < max_evaluate_depth)
    mask = np.where(np.logical_and(validity_mask, min_max_mask) > 0)
    o, g = o[mask], g[mask]
    src, tgt = 1000.0 * o, 1000.0 * g
    isrc, itgt = 0.001 * o, 0.001 * g
    mae = np.mean(np.abs(tgt - src))
    rmse = np.sqrt(np.mean((tgt - src) ** 2))
    imae = np.mean(np.abs((1.0 / itgt) - (1.0 / isrc)))
    irmse = np.sqrt(np.mean(((1.0 / itgt) - (1.0 / isrc)) ** 2))
    return float(mae), float(rmse), float(imae), float(irmse)


# ------------------------------------------------------- "next" row f4: input pipeline
def load_inference_sample(image_path, sparse_depth_path, intrinsics_path, use_image_triplet=True):
    """datasets.KBNetInferenceDataset.__getitem__ (reference src/datasets.py:259-283): the image
    through PIL's `Image.open(path).convert('RGB')` un-normalised (data_utils.load_image,
    src/data_utils.py:58-85), the middle third of a triplet (load_image_triplet, src/datasets.py:22-46),
    the 16-bit depth PNG / 256 as 1 x H x W (data_utils.load_depth, src/data_utils.py:123-152), the
    intrinsics .npy as float32.  Returns numpy (image 3 x H x W, sparse_depth 1 x H x W, intrinsics 3 x 3)."""
    import numpy as np
    from PIL import Image
    image = np.asarray(Image.open(image_path).convert("RGB"), np.float32)
    image = np.transpose(image, (2, 0, 1))
    if use_image_triplet:
        _, image, _ = np.split(image, indices_or_sections=3, axis=-1)
    z = np.array(Image.open(sparse_depth_path), dtype=np.float32)
    z = z / 256.0
    z[z <= 0] = 0.0
    z = np.expand_dims(z, axis=0)
    intrinsics = np.load(intrinsics_path).astype(np.float32)
    return image.astype(np.float32), z.astype(np.float32), intrinsics
