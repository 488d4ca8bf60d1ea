#!/usr/bin/env python3
"""conv_fused of KB2-4 (1x1 stride 2 over cat[image, xyz, fused]): the in-kernel backprojection source against a plain
3-channel tensor in its place -- what the synthesis costs (GPU box).  usage: fused1x1_bench.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for name, ci, cd, cf, h, w in [("kb2", 48, 16, 48, 176, 608), ("kb3", 96, 32, 96, 88, 304), ("kb4", 192, 64, 192, 44, 152)]:
    image = torch.randn(B, ci, h, w, generator=g).to(dev)
    depth = torch.rand(B, cd, h, w, generator=g).to(dev)
    fused = torch.randn(B, cf, h, w, generator=g).to(dev)
    fake = torch.randn(B, 3, h, w, generator=g).to(dev)
    proj = (torch.randn(1, cd, 1, 1, generator=g) / cd).to(dev)
    kinv = torch.eye(3).repeat(B, 1, 1).to(dev)
    cout = 2 * ci
    wt = (torch.randn(cout, ci + 3 + cf, 1, 1, generator=g) / (ci + cf) ** 0.5).to(dev)
    pw = kb.ops.pack_conv_weight(wt, 2)
    out = torch.empty(B, cout, h // 2, w // 2, device=dev)
    res = {}
    for tag, mid in (("xyz", kb.ops.xyz_src(depth, proj, kinv)), ("tensor", kb.ops.tensor_src(fake))):
        srcs = [kb.ops.tensor_src(image), mid, kb.ops.tensor_src(fused)]
        f = lambda: kb.ops.conv2d(srcs, pw, B, cout, 1, 2, h, w, out, negative_slope=0.2)
        with kb.ops.autotune():
            f()
        for _ in range(3): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        res[tag] = s.elapsed_time(e) * 100
    print(f"{name}: in-kernel xyz {res['xyz']:.1f} us, 3-channel tensor in its place {res['tensor']:.1f} us")
