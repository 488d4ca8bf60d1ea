#!/usr/bin/env python3
"""How far is the fp32 oracle itself from an fp64 evaluation of the same network?  (CPU only; lives under tests/
because it runs the oracle.)  The HIP path's distance to the fp32 oracle (tests/test_hip_parity.py seed sweep:
up to 5.7e-5 on KITTI) has to be read against this number: two different fp32 evaluation orders of a 35-conv
network cannot agree better than each of them agrees with the exact result.
usage: python tests/analysis/oracle_fp64_distance.py [seed ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kbnet_amd as kb
from oracle import kbnet_oracle as orc

torch.set_num_threads(min(16, os.cpu_count() or 1))
seeds = [int(a) for a in sys.argv[1:]] or [0, 3, 7, 11, 19]
for preset, shape in (("kitti", (352, 1216)), ("void", (480, 640))):
    cfg = kb.PRESETS[preset]()
    for seed in seeds:
        sds = kb.synthetic.make_state_dicts(cfg, seed=seed, gain=kb.synthetic.PARITY_GAIN[preset])
        frames = kb.synthetic.make_frames(1, *shape, preset, seed=1 + seed, jitter_intrinsics=0.1)
        args = (cfg.min_pools, cfg.max_pools, cfg.min_predict_depth, cfg.max_predict_depth)
        t = time.time()
        ref32 = orc.kbnet_forward(*frames, *sds, *args)
        torch.set_default_dtype(torch.float64)      # the pixel grid follows the default dtype (reference quirk Q8)
        try:
            ref64 = orc.kbnet_forward(*[f.double() for f in frames], *[{k: v.double() for k, v in sd.items()} for sd in sds], *args)
        finally:
            torch.set_default_dtype(torch.float32)
        err = (ref32.double() - ref64).abs() / ref64.abs()
        print(f"{preset:6s} seed {seed:2d}: fp32 oracle vs fp64: max rel {float(err.max()):.3e}  mean {float(err.mean()):.3e}  ({time.time() - t:.0f} s)",
              flush=True)
