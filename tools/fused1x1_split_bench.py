#!/usr/bin/env python3
"""conv_fused of KB2-KB4 (1x1 stride 2 over cat[image, xyz, fused]) at KITTI shapes: split-operand kernel (+ the xyz
kernel) next to the fp32 conv kernel with in-kernel xyz synthesis (GPU box).  usage: fused1x1_split_bench.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timed(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 100


for name, ci, cf, cd, cout, h, w in [("kb2_fused", 48, 48, 16, 96, 176, 608), ("kb3_fused", 96, 96, 32, 192, 88, 304),
                                     ("kb4_fused", 192, 192, 64, 384, 44, 152)]:
    oh, ow = (h + 1) // 2, (w + 1) // 2
    lrelu = torch.nn.functional.leaky_relu
    image = lrelu(torch.randn(B, ci, h, w, generator=g), 0.2).to(dev)
    fused = lrelu(torch.randn(B, cf, h, w, generator=g), 0.2).to(dev)
    depth = lrelu(torch.randn(B, cd, h, w, generator=g), 0.2).to(dev)
    proj = (torch.randn(1, cd, 1, 1, generator=g) / cd ** 0.5).to(dev)
    kinv = kb.ops.intrinsics_inverse(torch.tensor([[[700.0, 0.0, w / 2.0], [0.0, 700.0, h / 2.0], [0.0, 0.0, 1.0]]]).repeat(B, 1, 1).to(dev))
    wt = (torch.randn(cout, ci + 3 + cf, 1, 1, generator=g) / (ci + 3 + cf) ** 0.5).to(dev)
    srcs = [kb.ops.tensor_src(image), kb.ops.tensor_src(fused)]
    ps = kb.ops.pack_conv1x1s2_split_weight(wt, ci)
    out_s = torch.empty(B, cout, oh, ow, device=dev)
    out_f = torch.empty_like(out_s)
    xyz = kb.ops.kb_xyz_s2(depth, proj, kinv, 0.2)
    fx = lambda: kb.ops.kb_xyz_s2(depth, proj, kinv, 0.2, out=xyz)
    fs = lambda: kb.ops.conv1x1s2_split(srcs, ps, xyz, B, cout, oh, ow, out_s, negative_slope=0.2)
    pf = kb.ops.pack_conv_weight(wt, 2)
    s32 = [kb.ops.tensor_src(image), kb.ops.xyz_src(depth, proj, kinv), kb.ops.tensor_src(fused)]
    ff = lambda: kb.ops.conv2d(s32, pf, B, cout, 1, 2, h, w, out_f, negative_slope=0.2)
    with kb.ops.autotune():
        ff()
    tx, ts, tf = timed(fx), timed(fs), timed(ff)
    gb = ((ci + cf) * (h // 2) * w + cout * oh * ow) * 4 * B / 1e9     # even rows of the inputs + the output
    print(f"{name:10s} xyz {tx:6.1f} us  split {ts:7.1f} us ({gb / ts * 1e3:5.2f} TB/s of even-row traffic)  | fp32 {tf:7.1f} us   "
          f"max |split - fp32| / max |fp32| = {float((out_s - out_f).abs().max() / out_f.abs().max()):.1e}")
