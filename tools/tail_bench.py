#!/usr/bin/env python3
"""conv_tail alone (deconv0's second conv on split operands + output0 + depth mapping) on KITTI 352x1216 (GPU box).
usage: tail_bench.py [batch] [reps]      TAIL_ONCE=1: five launches only (PMC runs)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
h, w, c = 352, 1216, 12
x = torch.randn(B, c, h, w, generator=g).to(dev)
wc = (torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5).to(dev)
wo = (torch.randn(1, c, 3, 3, generator=g) * 0.5).to(dev)
packed = kb.ops.pack_conv_tail_weight(wc)
out = torch.empty(B, 1, h, w, device=dev)
run = lambda: kb.ops.conv_tail(x, packed, wo, 1.5, 100.0, 0.2, out=out)
head = lambda: kb.ops.conv_head(x, wc, wo, 1.5, 100.0, 0.2, out=out)
for _ in range(5):
    run()
torch.cuda.synchronize()
if os.environ.get("TAIL_ONCE"):
    sys.exit(0)
for name, f in (("conv_tail (fp16 split)", run), ("conv_head (fp32 MFMA)", head)):
    for _ in range(3):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    print(f"{name} batch {B}: {s.elapsed_time(e) * 1e3 / reps:.1f} us")
