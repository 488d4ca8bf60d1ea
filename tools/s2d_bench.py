#!/usr/bin/env python3
"""Times kbn_s2d_forward alone (KITTI preset, batch 8) under the KBN_S2D_DEBUG ablation switches."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch, kbnet_amd as kb
    preset = os.environ.get("S2D_PRESET", "kitti")
    cfg = kb.PRESETS[preset]()
    h, w = (352, 1216) if preset == "kitti" else (480, 640)
    dev = torch.device("cuda:0")
    nb = int(os.environ.get("S2D_BATCH", "8"))
    _, sp, va, _ = kb.synthetic.make_frames(nb, h, w, preset, seed=1)
    x = torch.cat([sp, va], 1).to(dev)
    sd = kb.synthetic.make_state_dicts(cfg, seed=0)[0]
    ws = [sd[f"pool_convs.{i}.conv.weight"].to(dev) for i in range(3)]
    wc = sd["conv.conv.weight"].to(dev)
    f = lambda: kb.ops.s2d_forward(x, ws, wc, cfg.min_pools, cfg.max_pools)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(json.dumps({"us": round(us, 1), "GBps": round(nb * h * w * 40 / us / 1e3, 1), "batch": nb}))
    sys.exit(0)
cases = ((0, "full"), (16, "no z staging"), (1, "no vertical pass"), (2, "no horizontal pass"), (4, "no 1x1 chain"), (8, "no 3x3 conv"), (12, "no convs"), (31, "skeleton only"))
for dbg, tag in cases:
    r = subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, KBN_S2D_DEBUG=str(dbg)), capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(f"{tag:26s}", line[-1] if line else r.stderr[-300:], flush=True)
