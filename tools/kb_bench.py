#!/usr/bin/env python3
"""Times the KB blocks of the KITTI forward (batch KBN_BATCH, default 8) one by one, through kbn_kb_block_forward.
usage: kb_bench.py [level ...]   (levels 1-4; prints µs per block launch sequence, median of 5 blocks of 8)
KBN_NO_KB_PAIR=1 times the three-launch path."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kbnet_amd as kb

SHAPES = {1: (48, 16, 0, 48, 16, 352, 1216), 2: (48, 16, 48, 96, 32, 176, 608),
          3: (96, 32, 96, 192, 64, 88, 304), 4: (192, 64, 192, 384, 128, 44, 152)}


def run(level, batch):
    ci, cd, cf, fi, fd, h, w = SHAPES[level]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(level)
    blk = kb.modules.CalibratedBackprojectionBlock(ci, cd, ci + cf, fi, fd, fi, 1, 1, 1, "xavier_normal",
                                                   torch.nn.LeakyReLU(0.2)).to(dev)
    image = torch.randn(batch, ci, h, w, generator=g).to(dev)
    depth = torch.randn(batch, cd, h, w, generator=g).to(dev)
    fused = torch.randn(batch, cf, h, w, generator=g).to(dev) if cf else None
    kinv = kb.ops.intrinsics_inverse(torch.tensor([[[721.5, 0.0, w / 2.0], [0.0, 721.5, h / 2.0], [0.0, 0.0, 1.0]]]).repeat(batch, 1, 1).to(dev))
    oh, ow = (h + 1) // 2, (w + 1) // 2
    outs = [torch.empty(batch, c, oh, ow, device=dev) for c in (fi, fd, fi)]
    f = lambda: blk.run(image, depth, kinv, fused, *outs)
    for _ in range(6):
        f()
    torch.cuda.synchronize()
    samples = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(8):
            f()
        e.record()
        torch.cuda.synchronize()
        samples.append(s.elapsed_time(e) * 1e3 / 8)
    flops = 2.0 * batch * oh * ow * (9 * ci * fi + 9 * (cd + 3) * fd + (ci + 3 + cf) * fi)
    us = sorted(samples)[2]
    return {"level": level, "us": round(us, 1), "tflops": round(flops / us / 1e6, 1)}


if __name__ == "__main__":
    levels = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
    for lv in levels:
        print(json.dumps(run(lv, int(os.environ.get("KBN_BATCH", "8")))), flush=True)
