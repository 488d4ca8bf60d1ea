#!/usr/bin/env python3
"""kb1_front alone (conv0_image + KB1's conv_image / conv_fused in one launch) on KITTI 352x1216 frames (GPU box).
usage: front_bench.py [batch] [reps]      FRONT_ONCE=1: five launches only (PMC runs)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
h, w = 352, 1216
image = torch.rand(B, 3, h, w, generator=g).to(dev)
w0 = (torch.randn(48, 3, 3, 3, generator=g) / 27 ** 0.5).to(dev)
wi = (torch.randn(48, 48, 3, 3, generator=g) / 432 ** 0.5).to(dev)
wf = (torch.randn(48, 51, 1, 1, generator=g) / 51 ** 0.5).to(dev)
xyz = torch.randn(B, 3, h // 2, w // 2, generator=g).to(dev)
packed = kb.ops.pack_kb1_front_weight(w0, wi, wf)
oi = torch.empty(B, 48, h // 2, w // 2, device=dev)
of = torch.empty_like(oi)
stats = kb.ops.ActStats(B, dev)
a, b = stats.new(), stats.new()
run = lambda: kb.ops.kb1_front(image, packed, xyz, 48, 48, oi, of, 0.2, 0.2, a, b)
for _ in range(5):
    run()
torch.cuda.synchronize()
if os.environ.get("FRONT_ONCE"):
    sys.exit(0)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    run()
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / reps
tiles = B * (h // 2 // 8) * (-(-(w // 2) // 16))
mfma = tiles * 3 * (36 * 6 + 8 * 54)
print(f"kb1_front batch {B}: {us:.1f} us, {mfma * 16384 * 2 / us / 1e6:.0f} TFLOP/s issued fp16 MFMA ({mfma / 1e6:.1f} M MFMAs of 16x16x32)")
