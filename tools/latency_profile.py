#!/usr/bin/env python3
"""One KITTI frame, default form against KBNetModel.set_latency_mode(): per-launch times of an eager pass (HIP events) and the graph-replayed latency.
usage: latency_profile.py [batch]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kbnet_amd as kb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
cfg = kb.kitti_config()
m = kb.modules.KBNetModel.from_config(cfg, dev)
m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN["kitti"]))
fr = [f.to(dev) for f in kb.synthetic.make_frames(n, 352, 1216, "kitti", seed=1)]
for mode in (False, True):
    m.set_latency_mode(mode)
    for _ in range(3): m.forward(*fr)
    torch.cuda.synchronize()
    kb.ops.PROFILE = []
    for _ in range(10): m.forward(*fr)
    torch.cuda.synchronize()
    prof, kb.ops.PROFILE = kb.ops.PROFILE, None
    per = len(prof) // 10
    rows = [(p[0], sum(q[5].elapsed_time(q[6]) for q in prof[i::per]) * 100) for i, p in enumerate(prof[:per])]
    g = m.capture(*fr)
    for _ in range(20): g(*g.static_in)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): g(*g.static_in)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 5
    print(f"latency_mode {mode}: graph replay {ms:.3f} ms per batch of {n}; eager launches (us): total {sum(r[1] for r in rows):.0f}")
    print("   " + "  ".join(f"{a}:{b:.0f}" for a, b in rows))
