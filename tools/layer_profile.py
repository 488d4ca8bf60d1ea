#!/usr/bin/env python3
"""Per-launch durations of one eager KITTI 352x1216 forward at batch B (default 32), averaged over 5 passes, HIP events on
the launch stream (GPU box).  usage: layer_profile.py [batch] [name-substring]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pat = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
cfg = kb.PRESETS["kitti"]()
m = kb.modules.KBNetModel.from_config(cfg, dev)
m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN["kitti"]))
fr = [f.to(dev) for f in kb.synthetic.make_frames(B, 352, 1216, "kitti", seed=1)]
# A/B switches of the host mirror: LP_KB1_SPLIT=1, LP_NARROW_UP=1 (16-filter split tiles for deconv0's up-conv), LP_FUSED_MIN=<filters>
# (conv_fused on split operands from this width on)
if os.environ.get("LP_KB1_SPLIT"):   # KB1's conv_image (48 filters) on the stride-2 split kernel instead of the fused fp32 KB kernel
    m.encoder.calibrated_backprojection1.split_image = True
for mod in m.modules():
    for sub in mod.modules():
        if isinstance(sub, kb.modules.Conv2d):
            if os.environ.get("LP_NARROW_UP"):
                sub.split_narrow_up = True
            if os.environ.get("LP_FUSED_MIN"):
                sub.split_fused_min_filters = int(os.environ["LP_FUSED_MIN"])
with kb.ops.autotune():
    m.forward(*fr)
for _ in range(2):
    m.forward(*fr)
torch.cuda.synchronize()
kb.ops.PROFILE = []
for _ in range(5):
    m.forward(*fr)
torch.cuda.synchronize()
prof, kb.ops.PROFILE = kb.ops.PROFILE, None
per = len(prof) // 5
tot = 0.0
by = {}
for i in range(per):
    name = prof[i][0]
    us = sum(prof[r * per + i][-2].elapsed_time(prof[r * per + i][-1]) for r in range(5)) * 1e3 / 5
    tot += us
    by[name] = by.get(name, 0.0) + us
    if pat and pat in name:
        print(f"{us:8.1f} us  {name}")
print("  ".join(f"{k} {v:.0f}" for k, v in by.items()))
print(f"sum {tot:.1f} us -> {B / tot * 1e6:.0f} frames/s eager")
