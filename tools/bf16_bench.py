#!/usr/bin/env python3
"""Per-layer timing of the throughput-only bf16 decoder convs (kbn_conv3x3_bf16_forward), KITTI shapes (GPU box).
usage: bf16_bench.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
# name: (source channels, cout, H, W (output), up2x)
LAYERS = [("deconv4_up", (512,), 256, 22, 76, True), ("deconv4_conv", (256, 512), 256, 22, 76, False),
          ("deconv3_up", (256,), 128, 44, 152, True), ("deconv3_conv", (128, 256), 128, 44, 152, False),
          ("deconv2_up", (128,), 128, 88, 304, True), ("deconv2_conv", (128, 128), 128, 88, 304, False),
          ("deconv1_up", (128,), 64, 176, 608, True), ("deconv1_conv", (64, 64), 64, 176, 608, False),
          ("deconv0_up", (64,), 12, 352, 1216, True)]
g = torch.Generator().manual_seed(0)
tot = 0.0
for name, cins, cout, h, w, up in LAYERS:
    sh, sw = (h // 2, w // 2) if up else (h, w)
    xs = [torch.randn(B, c, sh, sw, generator=g).to(dev) for c in cins]
    cin = sum(cins)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
    pw = kb.ops.pack_conv3x3_bf16_weight(wt)
    out = torch.empty(B, cout, h, w, device=dev)
    srcs = [kb.ops.tensor_src(x) for x in xs]
    f = lambda: kb.ops.conv3x3_bf16(srcs, pw, B, cout, h, w, out, up2x=up)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    byts = 4.0 * B * (cin * sh * sw + cout * h * w)
    flops = 2.0 * B * h * w * cin * 9 * cout
    tot += us
    print(f"{name:14s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {byts / us / 1e3:7.1f} GB/s algorithmic", flush=True)
print(f"sum {tot:.1f} us per {B} frames")
