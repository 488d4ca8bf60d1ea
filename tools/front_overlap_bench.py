#!/usr/bin/env python3
"""Experiment: inside each sub-batch branch of the captured graph, run S2D -> conv0_depth (vector-ALU / LDS bound) on
a side stream while conv0_image (HBM-write bound, independent of S2D) runs on the branch's own stream.
Prints frames/s of the default 2-branch graph and of the variant, and whether the outputs are identical."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
from kbnet_amd import ops
dev = torch.device("cuda:0")
cfg = kb.kitti_config()
m = kb.modules.KBNetModel.from_config(cfg, dev)
m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN["kitti"]))
fr = [f.to(dev) for f in kb.synthetic.make_frames(8, 352, 1216, "kitti", seed=1)]


def fwd_overlap(image, sparse, valid, k, side, out, x, fork):
    """x = cat[sparse, valid] and `fork` (an event after it) come from the capture's origin stream: the side stream
    forks from there, not from the branch stream (nested forks crash hipStreamEndCapture on this ROCm)."""
    enc, s2dm = m.encoder, m.sparse_to_dense_pool
    n, _, h, w = image.shape
    main = torch.cuda.current_stream()
    s2d = torch.empty((n, s2dm.conv.out_channels, h, w), device=dev)
    img0 = torch.empty((n, enc.conv0_image.out_channels, h, w), device=dev)
    dep0 = torch.empty((n, enc.conv0_depth.out_channels, h, w), device=dev)
    side.wait_event(fork)
    with torch.cuda.stream(side):
        ops.s2d_forward(x, [c.conv.weight for c in s2dm.pool_convs], s2dm.conv.conv.weight,
                        s2dm.min_pool_sizes, s2dm.max_pool_sizes, s2dm._slope, out=s2d)
        enc.conv0_depth.run([ops.tensor_src(s2d)], n, h, w, out=dep0)
        done = side.record_event()
    enc.conv0_image.run([ops.tensor_src(image)], n, h, w, out=img0)
    main.wait_event(done)
    enc.conv0_image.forward = lambda _x: img0       # the encoder continues from the two conv0 outputs
    enc.conv0_depth.forward = lambda _x: dep0
    try:
        latent, skips = enc(image, s2d, k)
    finally:
        del enc.conv0_image.forward, enc.conv0_depth.forward
    feats = m.decoder.features(latent, skips, (h, w))
    return ops.depth_head(feats, m.decoder.output0.conv.weight, m.min_predict_depth, m.max_predict_depth, out=out)


def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return 8 * reps / (time.perf_counter() - t)


base = m.capture(*fr)                       # default: 2 concurrent sub-batch branches
ref = base(*fr).clone()
parts = [[f[i * 4:(i + 1) * 4].contiguous() for f in fr] for i in range(2)]
out = torch.empty_like(ref)
sides = [torch.cuda.Stream() for _ in range(2)]
sB = torch.cuda.Stream()
for p in parts:                              # warm the eager path (shapes are tuned already)
    x = torch.cat([p[1], p[2]], dim=1)
    fwd_overlap(*p, sides[0], out[0:4], x, torch.cuda.current_stream().record_event())
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    xs = [torch.cat([p[1], p[2]], dim=1) for p in parts]
    fork = cur.record_event()
    sB.wait_event(fork)
    with torch.cuda.stream(sB):
        fwd_overlap(*parts[1], sides[1], out[4:8], xs[1], fork)
    fwd_overlap(*parts[0], sides[0], out[0:4], xs[0], fork)
    cur.wait_stream(sB)
    for s_ in sides:
        cur.wait_stream(s_)
for rep in range(3):
    print("default 2-branch graph            : %.1f frames/s" % timeit(lambda: base(*fr)))
    print("S2D/conv0_depth || conv0_image    : %.1f frames/s  same bits: %s" % (timeit(g.replay), torch.equal(out, ref)))
