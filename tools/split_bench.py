#!/usr/bin/env python3
"""Per-layer timing and fp64-referenced error of the split-operand convs (kbn_conv3x3_split_forward: fp32 products as three
fp16 MFMAs) next to the fp32-MFMA kernels the same layers run otherwise (Winograd / 9-product up-conv), KITTI shapes (GPU box).
usage: split_bench.py [batch]      SPLIT_ZERO=1: zero operands (DVFS check)   SPLIT_AMAG=x: activation magnitude"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
AMAG = float(os.environ.get("SPLIT_AMAG", "1"))
# name: (source channels, cout, H, W (output), up2x)
LAYERS = [("deconv4_up", (512,), 256, 22, 76, True), ("deconv4_conv", (256, 512), 256, 22, 76, False),
          ("deconv3_up", (256,), 128, 44, 152, True), ("deconv3_conv", (128, 256), 128, 44, 152, False),
          ("deconv2_up", (128,), 128, 88, 304, True), ("deconv2_conv", (128, 128), 128, 88, 304, False),
          ("deconv1_up", (128,), 64, 176, 608, True), ("deconv1_conv", (64, 64), 64, 176, 608, False),
          ("deconv0_up", (64,), 12, 352, 1216, True),
          ("kb2_image", (48,), 96, 88, 304, "s2"), ("kb3_image", (96,), 192, 44, 152, "s2"), ("kb4_image", (192,), 384, 22, 76, "s2")]
ONLY = os.environ.get("SPLIT_ONLY")           # one layer, split kernel only (PMC runs)
if ONLY:
    LAYERS = [l for l in LAYERS if l[0] == ONLY]
g = torch.Generator().manual_seed(0)


def timed(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 100


tot = [0.0, 0.0]
for name, cins, cout, h, w, up in LAYERS:
    stride = 2 if up == "s2" else 1
    up = up is True
    sh, sw = (h // 2, w // 2) if up else ((2 * h, 2 * w) if stride == 2 else (h, w))
    xs = [(AMAG * torch.nn.functional.leaky_relu(torch.randn(B, c, sh, sw, generator=g), 0.2)).to(dev) for c in cins]
    cin = sum(cins)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
    if os.environ.get("SPLIT_ZERO"):      # DVFS check: zero operands draw less power (same instruction stream)
        xs = [torch.zeros_like(x) for x in xs]; wt = torch.zeros_like(wt)
    srcs = [kb.ops.tensor_src(x) for x in xs]
    out_s = torch.empty(B, cout, h, w, device=dev)
    out_f = torch.empty(B, cout, h, w, device=dev)
    ps = kb.ops.pack_conv3x3_split_weight(wt, stride=stride, folded_up2x=up)
    fs = lambda: kb.ops.conv3x3_split(srcs, ps, B, cout, h, w, out_s, up2x=up, negative_slope=0.2, stride=stride, folded_up2x=up)
    if up:
        pf = kb.ops.pack_upconv2x_weight(wt)
        ff = lambda: kb.ops.upconv2x(xs[0], pf, cout, out_f, 0.2)
    else:
        pf = kb.ops.pack_conv_weight(wt, stride)
        ff = lambda: kb.ops.conv2d(srcs, pf, B, cout, 3, stride, sh, sw, out_f, negative_slope=0.2)
    if ONLY:
        for _ in range(5): fs()
        torch.cuda.synchronize()
        continue
    with kb.ops.autotune():
        ff()
    assert fs() is not None
    us_s, us_f = timed(fs), timed(ff)
    xin = torch.cat([x[:1] for x in xs], 1).double().cpu()
    if up:
        xin = torch.nn.functional.interpolate(xin, size=(h, w), mode="nearest")
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xin, wt.double().cpu(), stride=stride, padding=1), 0.2)
    rms = float(ref.pow(2).mean().sqrt()) or 1.0
    es, ef = (out_s[:1].cpu().double() - ref).abs(), (out_f[:1].cpu().double() - ref).abs()
    flops = 2.0 * B * h * w * cin * 9 * cout
    tot[0] += us_s; tot[1] += us_f
    print(f"{name:13s} split {us_s:7.1f} us {flops / us_s / 1e6:6.1f} TFLOP/s (err vs fp64: max {float(es.max()) / rms:.2e} rms {float(es.pow(2).mean().sqrt()) / rms:.2e})"
          f" | fp32 MFMA {us_f:7.1f} us {flops / us_f / 1e6:6.1f} TFLOP/s (max {float(ef.max()) / rms:.2e} rms {float(ef.pow(2).mean().sqrt()) / rms:.2e})", flush=True)
print(f"sum: split {tot[0]:.1f} us, fp32 MFMA kernels {tot[1]:.1f} us per {B} frames")
