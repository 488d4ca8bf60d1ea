"""CPU: the oracle restatement reproduces the reference's captured outputs bit-for-bit."""
import pytest
import torch

from conftest import load_golden
from oracle import kbnet_oracle as orc

import kbnet_amd as kb


@pytest.mark.parametrize("name", [f"s2d_{p}_{i}" for p in ("kitti", "void") for i in range(3)])
def test_s2d_matches_reference(name):
    g = load_golden(name)
    pyr, out = orc.sparse_to_dense_pool(g["x"], g["weights"], list(g["min_pool_sizes"]),
                                        list(g["max_pool_sizes"]), return_pyramid=True)
    assert torch.equal(pyr, g["pyramid"])
    assert torch.equal(out, g["out"])


def test_s2d_sentinel_quirk():
    """Q6: a depth of exactly 999 (and a lone depth above it) min-pools to 0."""
    g = load_golden("s2d_kitti_0")
    pyr = g["pyramid"]
    assert g["x"][0, 0, 10, 12] == 999.0 and g["x"][0, 0, 10, 14] == 1500.0
    assert pyr[0, 0, 10, 12] == 0.0          # k=5 window holds only 999 and 1500
    assert pyr[0, 5, 10, 12] == 1500.0       # max pool sees it


@pytest.mark.parametrize("name", ["coords_kitti", "coords_nyu", "coords_odd"])
def test_coordinates_match_reference(name):
    g = load_golden(name)
    k, h, w = g["intrinsics"], int(g["height"]), int(g["width"])
    h1, w1 = (h + 1) // 2, (w + 1) // 2
    hl, wl = h, w
    for lvl in range(4):
        kl = k if lvl == 0 else orc.scale_intrinsics(k, h, w, h1, w1)
        c = orc.camera_coordinates(kl, hl, wl)
        assert torch.equal(c, g[f"coordinates{lvl}"]), lvl
        assert torch.all(c[:, 2] == 1.0)  # Q9
        hl, wl = (hl + 1) // 2, (wl + 1) // 2


@pytest.mark.parametrize("name", ["kb_nofused", "kb_fused", "kb_odd", "kb_stacked", "kb_stacked_odd"])
def test_kb_block_matches_reference(name):
    """kb_stacked*: n_convolution_image / n_convolution_depth > 1 (reference src/net_utils.py:1311-1325)."""
    g = load_golden(name)
    ci, cd, cf = orc.kb_block(g["image"], g["depth"], g["coordinates"], g.get("fused"), g["weights"])
    assert torch.equal(ci, g["conv_image"])
    assert torch.equal(cd, g["conv_depth"])
    assert torch.equal(cf, g["conv_fused"])


def test_encoder_with_stacked_convolutions_matches_reference():
    """enc_stacked: networks.KBNetEncoder with n_convolutions_image / _depth > 1 (src/networks.py:52-299), KB layers at levels 0 and 2,
    plain VGG levels with stacked convs elsewhere, an odd frame size -- the oracle follows the state_dict's conv_block.<i> keys."""
    g = load_golden("enc_stacked")
    levels = tuple(int(v) for v in g["resolutions_backprojection"])
    latent, skips = orc.encoder(g["image"], g["depth"], g["intrinsics"], g["weights"], resolutions_backprojection=levels)
    assert torch.equal(latent, g["latent"])
    for i, s_ in enumerate(skips):
        assert torch.equal(s_, g[f"skip{i + 1}"]), i


@pytest.mark.parametrize("name", ["dec_even", "dec_odd", "dec_transpose"])
def test_decoder_matches_reference(name):
    """dec_transpose: deconv_type='transpose' -- ConvTranspose2d(3, stride 2, padding 1, output_padding 1) up-sampling layers
    (reference src/net_utils.py:350-440); the oracle follows the state dict's `deconv.deconv.weight` keys."""
    g = load_golden(name)
    assert ("deconv4.deconv.deconv.weight" in g["weights"]) == (name == "dec_transpose")
    skips = [g[f"skip{i}"] for i in range(1, 5)]
    out = orc.decoder(g["latent"], skips, tuple(int(v) for v in g["shape"]), g["weights"])
    assert torch.equal(out, g["logits"])


def golden_config(g):
    """The KBNetConfig a fwd_* fixture was generated with (narrow widths; recorded widths / KB levels where they differ)."""
    import dataclasses
    cfg = kb.PRESETS[str(g["preset"])]().narrow()
    if "resolutions_backprojection" in g:
        cfg = dataclasses.replace(cfg, resolutions_backprojection=tuple(int(v) for v in g["resolutions_backprojection"]))
    if "n_filters_encoder_image" in g:
        cfg = dataclasses.replace(cfg, n_filters_encoder_image=tuple(int(v) for v in g["n_filters_encoder_image"]),
                                  n_filters_encoder_depth=tuple(int(v) for v in g["n_filters_encoder_depth"]))
    if "deconv_type" in g:
        cfg = dataclasses.replace(cfg, deconv_type=str(g["deconv_type"]))
    if "activation_func" in g:
        cfg = dataclasses.replace(cfg, activation_func=str(g["activation_func"]))
    return cfg


@pytest.mark.parametrize("name", ["fwd_kitti", "fwd_void", "fwd_odd", "fwd_kb012", "fwd_kb02", "fwd_kb01234", "fwd_kb01234_odd",
                                  "fwd_transpose", "fwd_transpose_void", "fwd_act_relu", "fwd_act_elu", "fwd_act_sigmoid", "fwd_act_linear",
                                  "fwd_act_elu_kb012_transpose"])
def test_forward_matches_reference(name):
    """fwd_kb012 / fwd_kb02: encoders with KB layers at levels [0, 1, 2] / [0, 2] only; fwd_kb01234*: a KB layer at resolution 4
    too, where the reference calls calibrated_backprojection4 a second time (src/networks.py:499-517, quirk Q3); fwd_transpose*:
    deconv_type='transpose'; fwd_act_*: activation_func relu / elu / sigmoid / linear on every layer (reference
    src/net_utils.py:23-45; orc.activation_slope maps the string)."""
    g = load_golden(name)
    cfg = golden_config(g)
    levels = tuple(int(v) for v in g["resolutions_backprojection"]) if "resolutions_backprojection" in g else (0, 1, 2, 3)
    out = orc.kbnet_forward(g["image"], g["sparse_depth"], g["validity_map"], g["intrinsics"],
                            g["s2d"], g["encoder"], g["decoder"],
                            cfg.min_pools, cfg.max_pools, cfg.min_predict_depth, cfg.max_predict_depth,
                            resolutions_backprojection=levels, slope=orc.activation_slope(cfg.activation_func))
    assert torch.equal(out, g["output_depth"])
    assert out.min() >= cfg.min_predict_depth * 0.98 and out.max() <= cfg.max_predict_depth * 1.001


def test_outlier_removal_matches_reference():
    """validity map + OutlierRemoval (batch-global fill value) + image / 255."""
    g = load_golden("pre_outlier")
    fs, fv = orc.validity_and_outlier_removal(g["sparse_depth"], int(g["kernel_size"]), float(g["threshold"]))
    assert torch.equal(fv, g["filtered_validity_map"])
    assert torch.equal(fs, g["filtered_sparse_depth"])
    assert float((g["validity_map"] - g["filtered_validity_map"]).sum()) >= 2  # the planted outliers are dropped
    assert torch.equal(g["image"] / 255.0, g["image_normalized"])


def test_evaluation_metrics_match_reference():
    """MAE / RMSE / iMAE / iRMSE with the reference's masks and scaling."""
    g = load_golden("eval_metrics")
    m = orc.evaluation_metrics(g["output_depth"].numpy(), g["ground_truth"].numpy(), g["validity_map"].numpy(),
                               float(g["min_evaluate_depth"]), float(g["max_evaluate_depth"]))
    import numpy as np
    assert np.allclose(m, g["metrics"].numpy() if hasattr(g["metrics"], "numpy") else g["metrics"], rtol=1e-6)
