"""Hyper-parameters of the KBNet inference path and the two shipped presets.

The values mirror what the reference's run scripts pass on the command line
(reference `bash/kitti/run_kbnet_kitti_validation.sh:15-27`,
`bash/void/run_kbnet_void1500.sh:15-27`) and the defaults in
`src/global_constants.py:17-85`; the argument names follow
`KBNetModel.__init__` (`src/kbnet_model.py:63-79`).
"""

from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class KBNetConfig:
    name: str = "kitti"
    input_channels_image: int = 3
    input_channels_depth: int = 2
    min_pool_sizes_sparse_to_dense_pool: Tuple[int, ...] = (5, 7, 9, 11, 13)
    max_pool_sizes_sparse_to_dense_pool: Tuple[int, ...] = (15, 17)
    n_convolution_sparse_to_dense_pool: int = 3
    n_filter_sparse_to_dense_pool: int = 8
    n_filters_encoder_image: Tuple[int, ...] = (48, 96, 192, 384, 384)
    n_filters_encoder_depth: Tuple[int, ...] = (16, 32, 64, 128, 128)
    resolutions_backprojection: Tuple[int, ...] = (0, 1, 2, 3)
    n_filters_decoder: Tuple[int, ...] = (256, 128, 128, 64, 12)
    deconv_type: str = "up"
    weight_initializer: str = "xavier_normal"
    activation_func: str = "leaky_relu"
    min_predict_depth: float = 1.5
    max_predict_depth: float = 100.0

    # ---- derived ---------------------------------------------------------
    @property
    def min_pools(self) -> List[int]:
        # sizes <= 1 are dropped (reference src/networks.py:2112-2118)
        return [s for s in self.min_pool_sizes_sparse_to_dense_pool if s > 1]

    @property
    def max_pools(self) -> List[int]:
        return [s for s in self.max_pool_sizes_sparse_to_dense_pool if s > 1]

    @property
    def n_filters_encoder_fused(self) -> Tuple[int, ...]:
        # reference src/kbnet_model.py:111
        return tuple(self.n_filters_encoder_image)

    @property
    def n_skips(self) -> List[int]:
        # reference src/kbnet_model.py:99-105
        enc = [i + z for i, z in zip(self.n_filters_encoder_image, self.n_filters_encoder_depth)]
        return enc[:-1][::-1] + [0]

    def narrow(self, name=None) -> "KBNetConfig":
        """Same topology with thin channels: used for small golden fixtures."""
        return replace(self, name=name or (self.name + "_narrow"),
                       n_filters_encoder_image=(8, 16, 32, 64, 64),
                       n_filters_encoder_depth=(4, 8, 16, 32, 32),
                       n_filters_decoder=(32, 16, 16, 8, 4))


def kitti_config() -> KBNetConfig:
    return KBNetConfig()


def void_config() -> KBNetConfig:
    return KBNetConfig(name="void",
                       min_pool_sizes_sparse_to_dense_pool=(15, 17),
                       max_pool_sizes_sparse_to_dense_pool=(23, 27, 29),
                       min_predict_depth=0.1, max_predict_depth=8.0)


PRESETS = {"kitti": kitti_config, "void": void_config, "nyu_v2": void_config}


# ---------------------------------------------------------------- parameter map
def s2d_param_shapes(cfg: KBNetConfig) -> Dict[str, Tuple[int, ...]]:
    """`state_dict` keys/shapes of `networks.SparseToDensePool` (reference
    src/networks.py:2136-2166)."""
    shapes = {}
    cin = len(cfg.min_pools) + len(cfg.max_pools)
    f = cfg.n_filter_sparse_to_dense_pool
    for i in range(cfg.n_convolution_sparse_to_dense_pool):
        shapes[f"pool_convs.{i}.conv.weight"] = (f, cin, 1, 1)
        cin = f
    shapes["conv.conv.weight"] = (f, f + cfg.input_channels_depth, 3, 3)
    return shapes


def encoder_param_shapes(cfg: KBNetConfig) -> Dict[str, Tuple[int, ...]]:
    """Keys/shapes of `networks.KBNetEncoder` (reference src/networks.py:52-299)."""
    fi, fd, ff = cfg.n_filters_encoder_image, cfg.n_filters_encoder_depth, cfg.n_filters_encoder_fused
    kb = cfg.resolutions_backprojection
    assert 0 in kb, "resolution 0 must use calibrated backprojection (the reference is undefined otherwise)"
    s = {}
    s["conv0_image.conv.weight"] = (fi[0], cfg.input_channels_image, 3, 3)
    s["conv0_depth.conv.weight"] = (fd[0], cfg.n_filter_sparse_to_dense_pool, 3, 3)

    def kb_block(idx, ci, cd, cf, n):
        p = f"calibrated_backprojection{idx}."
        s[p + "conv_image.conv_block.0.conv.weight"] = (fi[n], ci, 3, 3)
        s[p + "conv_depth.conv_block.0.conv.weight"] = (fd[n], cd + 3, 3, 3)
        s[p + "proj_depth.conv.weight"] = (1, cd, 1, 1)
        s[p + "conv_fused.conv.weight"] = (ff[n], cf + 3, 1, 1)

    # level 0: in_channels_fused = n_filters_image[0] (image only, fused=None)
    kb_block(1, fi[0], fd[0], fi[0], 0)
    for n in (1, 2, 3):
        if n in kb:
            cf = fi[n - 1] + ff[n - 1] if (n - 1) in kb else fi[n - 1]
            kb_block(n + 1, fi[n - 1], fd[n - 1], cf, n)
        else:
            s[f"conv{n + 1}_image.conv_block.0.conv.weight"] = (fi[n], fi[n - 1], 3, 3)
            s[f"conv{n + 1}_depth.conv_block.0.conv.weight"] = (fd[n], fd[n - 1], 3, 3)
    if 4 in kb:
        # the reference BUILDS calibrated_backprojection5 (src/networks.py:266-283) and then never calls it: its level-4 branch
        # re-uses calibrated_backprojection4 (:512, quirk Q3).  The parameters still exist in its state_dict.
        cf = fi[3] + ff[3] if 3 in kb else fi[3]
        kb_block(5, fi[3], fd[3], cf, 4)
    else:
        s["conv5_image.conv_block.0.conv.weight"] = (fi[4], fi[3], 3, 3)
        s["conv5_depth.conv_block.0.conv.weight"] = (fd[4], fd[3], 3, 3)
    return s


def decoder_param_shapes(cfg: KBNetConfig) -> Dict[str, Tuple[int, ...]]:
    """Keys/shapes of `networks.MultiScaleDecoder` as KBNet configures it
    (reference src/kbnet_model.py:127-137, src/networks.py:1634-1853)."""
    f = cfg.n_filters_decoder
    skips = cfg.n_skips
    cin = cfg.n_filters_encoder_image[-1] + cfg.n_filters_encoder_depth[-1]
    s = {}
    for i, name in enumerate(("deconv4", "deconv3", "deconv2", "deconv1", "deconv0")):
        if cfg.deconv_type == "transpose":   # TransposeConv2d.deconv = ConvTranspose2d: in x out x 3 x 3 (reference src/net_utils.py:383-390)
            s[f"{name}.deconv.deconv.weight"] = (cin, f[i], 3, 3)
        else:
            s[f"{name}.deconv.conv.conv.weight"] = (f[i], cin, 3, 3)
        s[f"{name}.conv.conv.weight"] = (f[i], f[i] + skips[i], 3, 3)
        cin = f[i]
    s["output0.conv.weight"] = (1, f[4], 3, 3)
    return s
