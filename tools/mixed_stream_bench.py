#!/usr/bin/env python3
"""SURVEY §8(d) config 5 on one GPU: an interleaved stream of VOID 480x640 and NYUv2 416x576 batches (VOID
preset) and KITTI 352x1216 batches (KITTI preset), per-frame intrinsics (+-10 %), two weight sets resident,
one captured graph per shape.  Prints frames/s of the interleaved stream next to the per-shape rates."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
models = {}
for preset in ("kitti", "void"):
    cfg = kb.PRESETS[preset]()
    m = kb.modules.KBNetModel.from_config(cfg, dev)
    m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=1.3))
    models[preset] = m
stream = [("void", "void", (480, 640)), ("void", "nyu_v2", (416, 576)), ("kitti", "kitti", (352, 1216))]
frames, replay = {}, {}
for preset, stats, shape in stream:
    frames[shape] = [f.to(dev) for f in kb.synthetic.make_frames(B, *shape, stats, seed=3, jitter_intrinsics=0.1)]
    replay[shape] = models[preset].capture(*frames[shape])

def rate(order, reps):
    for _, _, shape in order:
        replay[shape](*frames[shape])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        for _, _, shape in order:
            replay[shape](*frames[shape])
    torch.cuda.synchronize()
    return reps * len(order) * B / (time.perf_counter() - t)

for s in stream:
    print(f"{s[1]:7s} {s[2][0]}x{s[2][1]} alone : {rate([s], 20):8.1f} frames/s")
print(f"interleaved stream (batches of {B}, VOID/NYU/KITTI round robin): {rate(stream, 20):8.1f} frames/s")
