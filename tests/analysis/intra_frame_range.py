#!/usr/bin/env python3
"""Where does the intra-frame dynamic-range case (tests/test_hip_parity.py::test_forward_full_size_intra_frame_dynamic_range) lose its
digits?  Prints, for the shipped path and for KBN_NO_SPLIT=1 (every conv on the fp32 MFMAs), the worst pixel against fp64, its distance
to the nearest outlier pixel, the logit there and the fp32 oracle's error at the same pixel.   usage: intra_frame_range.py [outlier_scale]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kbnet_amd as kb
from oracle import kbnet_oracle as orc
import test_hip_parity as T

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1e4
dev = torch.device("cuda:0")
cfg = kb.kitti_config()
sds = kb.synthetic.make_state_dicts(cfg, seed=2, gain=kb.synthetic.PARITY_GAIN["kitti"])
image, sparse, valid, k = kb.synthetic.make_frames(1, 352, 1216, "kitti", seed=5, jitter_intrinsics=0.1)
g = torch.Generator().manual_seed(17)
image, sparse = image.clone(), sparse.clone()
ys, xs = torch.randint(0, 352, (12,), generator=g), torch.randint(0, 1216, (12,), generator=g)
image[0, :, ys, xs] = image[0, :, ys, xs] * scale + 50.0 * (scale > 1)
hit = valid[0, 0].nonzero()
pick = hit[torch.randperm(hit.shape[0], generator=g)[:40]]
sparse[0, 0, pick[:20, 0], pick[:20, 1]] = 0.004
sparse[0, 0, pick[20:, 0], pick[20:, 1]] = 655.0
fr = (image, sparse, valid, k)
ref = orc.kbnet_forward(*fr, *sds, cfg.min_pools, cfg.max_pools, cfg.min_predict_depth, cfg.max_predict_depth)
ref64 = T._fp64_forward(cfg, sds, fr)
eo = ((ref.double() - ref64).abs() / ref64.abs())[0, 0]
print(f"oracle vs fp64: max {float(eo.max()):.3e} at {divmod(int(eo.argmax()), 1216)}")
for knob in (None, "KBN_NO_SPLIT"):
    if knob:
        os.environ[knob] = "1"; kb.ops.reload_env()
    m = kb.modules.KBNetModel.from_config(cfg, dev)
    m.load_state_dicts(*sds)
    out, logits = m.forward(*[f.to(dev) for f in fr], return_logits=True)
    e = ((out.cpu().double() - ref64).abs() / ref64.abs())[0, 0]
    y, x = divmod(int(e.argmax()), 1216)
    d = float(((ys - y).float() ** 2 + (xs - x).float() ** 2).sqrt().min())
    print(f"{knob or 'shipped'}: HIP vs fp64 max {float(e.max()):.3e} at ({y}, {x}), nearest outlier {d:.1f} px away, logit {float(logits[0, 0, y, x]):.2f}, "
          f"oracle's error there {float(eo[y, x]):.3e}; 99.99th percentile HIP {float(e.flatten().kthvalue(int(0.9999 * e.numel())).values):.3e} "
          f"oracle {float(eo.flatten().kthvalue(int(0.9999 * eo.numel())).values):.3e}; pixels with HIP error > 2 x oracle max: {int((e > 2 * eo.max()).sum())}")
    if knob:
        os.environ.pop(knob); kb.ops.reload_env()
