#!/usr/bin/env python3
"""CPU experiment: how much error would Winograd F(2x2,3x3) in fp32 add to the decoder's 3x3
stride-1 convs?  Runs the oracle in fp64 (truth), in fp32 (the parity reference) and in fp32 with
the wide stride-1 3x3 convs replaced by an fp32 Winograd evaluation, on one full-size KITTI frame.
usage: winograd_error.py [min_cin]"""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import kbnet_amd as kb
from oracle import kbnet_oracle as orc

BT_ = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G_ = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT_ = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_conv3x3(x, w):
    n, c, h, wd = x.shape
    BT, G, AT = BT_.to(x.dtype), G_.to(x.dtype), AT_.to(x.dtype)
    he, we = h + (h & 1), wd + (wd & 1)
    xp = F.pad(x, (1, 1 + we - wd, 1, 1 + he - h))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # n c th tw 4 4
    v = torch.einsum("ij,nctujk,lk->nctuil", BT, d, BT)          # B^T d B
    u = torch.einsum("ij,ocjk,lk->ocil", G, w, G)                # G g G^T
    m = torch.einsum("ocil,nctuil->notuil", u, v)
    y = torch.einsum("ij,notujk,lk->notuil", AT, m, AT)          # n o th tw 2 2
    th, tw = y.shape[2], y.shape[3]
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], 2 * th, 2 * tw)
    return y[:, :, :h, :wd]


def run(frames, sds, cfg, dtype, wino_min_cin=None):
    fr = [f.to(dtype) for f in frames]
    sd = [{k: v.to(dtype) for k, v in d.items()} for d in sds]
    plain = orc.conv2d
    if wino_min_cin is not None:
        def conv2d(x, weight, stride=1, slope=orc.NEGATIVE_SLOPE):
            if weight.shape[-1] == 3 and stride == 1 and weight.shape[1] >= wino_min_cin:
                y = winograd_conv3x3(x, weight)
                return F.leaky_relu(y, negative_slope=slope) if slope is not None else y
            return plain(x, weight, stride, slope)
        orc.conv2d = conv2d
    default = torch.get_default_dtype()
    torch.set_default_dtype(dtype)   # the oracle's pixel grid follows the default dtype
    try:
        return orc.kbnet_forward(*fr, *sd, cfg.min_pools, cfg.max_pools, cfg.min_predict_depth, cfg.max_predict_depth)
    finally:
        orc.conv2d = plain
        torch.set_default_dtype(default)


def main():
    min_cin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (352, 1216)
    torch.set_num_threads(8)
    import bench
    cfg = kb.kitti_config()
    sds = kb.synthetic.make_state_dicts(cfg, seed=0, gain=bench.WEIGHT_GAIN)
    frames = kb.synthetic.make_frames(1, h, w, "kitti", seed=1)
    truth = run(frames, sds, cfg, torch.float64)
    f32 = run(frames, sds, cfg, torch.float32).double()
    wino = run(frames, sds, cfg, torch.float32, min_cin).double()
    rel = lambda a, b: float(((a - b).abs() / b.abs()).max())
    print(f"oracle fp32 vs fp64 truth : max elementwise rel {rel(f32, truth):.3e}")
    print(f"winograd fp32 vs fp64 truth: max elementwise rel {rel(wino, truth):.3e}")
    print(f"winograd fp32 vs oracle fp32: max elementwise rel {rel(wino, f32):.3e}  (the parity gate, 1e-4)")


if __name__ == "__main__":
    main()
