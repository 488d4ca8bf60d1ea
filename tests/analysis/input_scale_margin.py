#!/usr/bin/env python3
"""Off-distribution inputs (GPU box): the HIP forward, the fp32 oracle and an fp64 evaluation of the same network for the
input scales of test_forward_follows_input_scale_without_calibration.  Prints, per case and frame, max relative error
HIP vs fp32 oracle | HIP vs fp64 | fp32 oracle vs fp64 -- how much of a HIP-vs-oracle difference is the fp32 oracle's own
rounding.  KBN_NO_SPLIT=1 in the environment: the all-fp32-MFMA path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kbnet_amd as kb
from oracle import kbnet_oracle as orc
dev = torch.device("cuda:0")
torch.set_num_threads(16)
cfg = kb.kitti_config()
h, w = 352, 1216
sds = kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN["kitti"])
m = kb.modules.KBNetModel.from_config(cfg, dev)
m.load_state_dicts(*sds)
image, sparse, valid, k = kb.synthetic.make_frames(2, h, w, "kitti", seed=1, jitter_intrinsics=0.1)
vi, vs, vv, vk = kb.synthetic.make_frames(2, h, w, "void", seed=4, jitter_intrinsics=0.1)
cases = {
    "recorded frames": (image, sparse, valid, k),
    "image x 255 | depth x 10": (torch.cat([image[:1] * 255.0, image[1:]]), torch.cat([sparse[:1], sparse[1:] * 10.0]), valid, k),
    "depth x 0.1 | empty sparse map": (image, torch.cat([sparse[:1] * 0.1, torch.zeros_like(sparse[1:])]),
                                       torch.cat([valid[:1], torch.zeros_like(valid[1:])]), k),
    "VOID statistics": (vi, vs, vv, vk),
}
rel = lambda a, b: float(((a.double() - b.double()).abs() / b.double().abs()).max())
for name, fr in cases.items():
    out = m.forward(*[f.to(dev) for f in fr]).cpu()
    for i in range(2):
        one = [f[i:i + 1] for f in fr]
        ref = orc.kbnet_forward(*one, *sds, cfg.min_pools, cfg.max_pools, cfg.min_predict_depth, cfg.max_predict_depth)
        torch.set_default_dtype(torch.float64)
        try:
            ref64 = orc.kbnet_forward(*[f.double() for f in one], *[{k_: v.double() for k_, v in sd.items()} for sd in sds],
                                      cfg.min_pools, cfg.max_pools, cfg.min_predict_depth, cfg.max_predict_depth)
        finally:
            torch.set_default_dtype(torch.float32)
        print(f"{name:32s} frame {i}: HIP vs fp32 oracle {rel(out[i:i + 1], ref):.2e} | HIP vs fp64 {rel(out[i:i + 1], ref64):.2e} | "
              f"fp32 oracle vs fp64 {rel(ref, ref64):.2e}", flush=True)
