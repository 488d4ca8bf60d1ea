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
m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=1.3))
fr = [f.to(dev) for f in kb.synthetic.make_frames(B, 352, 1216, "kitti", seed=1)]
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
for i in range(per):
    name = prof[i][0]
    us = sum(prof[r * per + i][-2].elapsed_time(prof[r * per + i][-1]) for r in range(5)) * 1e3 / 5
    tot += us
    if pat in name:
        print(f"{us:8.1f} us  {name}")
print(f"sum {tot:.1f} us -> {B / tot * 1e6:.0f} frames/s eager")
