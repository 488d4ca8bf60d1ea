#!/usr/bin/env python3
"""Parity margin report (GPU box): max elementwise relative error of the HIP forward vs the oracle for the three full-size
presets and several seeds -- how far the 1e-4 gate is.  Lives under tests/ because it runs the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kbnet_amd as kb
from oracle import kbnet_oracle as orc
dev = torch.device("cuda:0")
torch.set_num_threads(16)
SEEDS = range(int(sys.argv[1])) if len(sys.argv) > 1 else (0, 3, 7)      # usage: parity_margin.py [number of seeds]
for preset, shape in (("kitti", (352, 1216)), ("void", (480, 640)), ("nyu_v2", (416, 576))):
    cfg = kb.PRESETS[preset]()
    worst = 0.0
    for seed in SEEDS:
        sds = kb.synthetic.make_state_dicts(cfg, seed=seed, gain=1.3 if preset == "kitti" else 1.45)
        frames = kb.synthetic.make_frames(1, *shape, preset, seed=1 + seed, jitter_intrinsics=0.1)
        m = kb.modules.KBNetModel.from_config(cfg, dev)
        m.load_state_dicts(*sds)
        out = m.forward(*[f.to(dev) for f in frames]).cpu()
        ref = orc.kbnet_forward(*frames, *sds, cfg.min_pools, cfg.max_pools, cfg.min_predict_depth, cfg.max_predict_depth)
        err = ((out - ref).abs() / ref.abs()).max()
        worst = max(worst, float(err))
        print(f"{preset:7s} seed {seed}: max rel err {float(err):.3e}", flush=True)
    print(f"{preset:7s} worst of {len(list(SEEDS))} seeds: {worst:.3e} (gate 1e-4)", flush=True)
