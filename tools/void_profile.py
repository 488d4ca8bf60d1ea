#!/usr/bin/env python3
"""Per-launch durations of one VOID 480x640 (VOID preset, batch 8) forward, eager, HIP events (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
dev = torch.device("cuda:0")
preset, shape = (sys.argv[1], (int(sys.argv[2]), int(sys.argv[3]))) if len(sys.argv) > 3 else ("void", (480, 640))
cfg = kb.PRESETS[preset]()
m = kb.modules.KBNetModel.from_config(cfg, dev)
m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN["void"]))
fr = [f.to(dev) for f in kb.synthetic.make_frames(8, *shape, preset, seed=1)]
for _ in range(3):
    m.forward(*fr)
torch.cuda.synchronize()
kb.ops.PROFILE = []
for _ in range(5):
    m.forward(*fr)
torch.cuda.synchronize()
prof, kb.ops.PROFILE = kb.ops.PROFILE, None
per = len(prof) // 5
tot = 0.0
for rec in prof[-per:]:
    name, work, s, e = rec[0], rec[1], rec[-2], rec[-1]
    us = s.elapsed_time(e) * 1e3
    tot += us
    print(f"{us:8.1f} us  {work / us / 1e6:7.1f} TF-equiv  {name}")
print(f"sum {tot:.1f} us -> {8 / tot * 1e6:.0f} frames/s eager")
