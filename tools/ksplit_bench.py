#!/usr/bin/env python3
"""One-frame launches of the decoder's / encoder's low-resolution split convs with and without split-K (GPU box).  usage: ksplit_bench.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kbnet_amd as kb
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for name, cins, cout, hw, stride in (("deconv4 conv", (256, 512), 256, (22, 76), 1), ("deconv3 conv", (128, 256), 128, (44, 152), 1),
                                     ("deconv2 conv", (128, 128), 128, (88, 304), 1), ("deconv1 conv", (64, 64), 64, (176, 608), 1),
                                     ("deconv4 upconv", (384,), 256, (22, 76), "up"), ("deconv3 upconv", (256,), 128, (44, 152), "up"), ("deconv2 upconv", (128,), 128, (88, 304), "up"),
                                     ("KB3 conv_image", (96,), 192, (44, 152), 2), ("KB4 conv_image", (192,), 384, (22, 76), 2), ("conv5", (384,), 384, (11, 38), 2)):
    for n in (1, 4):
        h, w = hw
        up = stride == "up"
        stride = 1 if up else stride
        sh, sw = (h // 2, w // 2) if up else ((2 * h, 2 * w) if stride == 2 else (h, w))
        xs = [torch.nn.functional.leaky_relu(torch.randn(n, c, sh, sw, generator=g), 0.2).to(dev) for c in cins]
        cin = sum(cins)
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
        stats = kb.ops.ActStats(n, dev)
        srcs = [kb.ops.tensor_src(x, "x", stats.measure(x)) for x in xs]
        packed = kb.ops.pack_conv3x3_split_weight(wt, stride=stride, folded_up2x=up)
        out = torch.empty(n, cout, h, w, device=dev)
        auto = kb.ops.ksplit_for(cin, cout, h, w, stride, up2x=up)
        row = []
        for ks in sorted({1, 2, 3, 4, 8, auto}):
            if ks > cin // 32 and ks != 1:
                continue
            f = lambda: kb.ops.conv3x3_split(srcs, packed, n, cout, h, w, out, negative_slope=0.2, stride=stride, ksplit=ks, up2x=up, folded_up2x=up)
            try:
                for _ in range(3): f()
            except kb.KbnError:
                continue
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): f()
            e.record(); torch.cuda.synchronize()
            row.append(f"ks {ks}{'*' if ks == auto else ''}: {s.elapsed_time(e) * 50:.1f} us")
        print(f"{name:16s} n {n}  " + "   ".join(row), flush=True)
