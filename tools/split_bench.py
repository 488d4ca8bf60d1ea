#!/usr/bin/env python3
"""Experiment: does running the two halves of a batch as two concurrent branches (two streams, one captured
graph) beat one batch-8 forward?  Kernel tails of one half could overlap the other half's kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
dev = torch.device("cuda:0")
cfg = kb.kitti_config()
m = kb.modules.KBNetModel.from_config(cfg, dev)
m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=1.3))
fr = [f.to(dev) for f in kb.synthetic.make_frames(8, 352, 1216, "kitti", seed=1)]
def branches(sizes):
    nb = len(sizes)
    offs = [sum(sizes[:i]) for i in range(nb + 1)]
    parts = [[f[offs[i]:offs[i + 1]].contiguous() for f in fr] for i in range(nb)]
    for h in parts:           # tune the smaller batch shapes outside any capture
        for _ in range(2):
            m.forward(*h)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(nb)]
    g = torch.cuda.CUDAGraph()
    outs = [None] * nb
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs[i] = m.forward(*parts[i])
        for s in streams:
            cur.wait_stream(s)
    return g, outs

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return 8 * reps / (time.perf_counter() - t)

for _ in range(2):
    m.forward(*fr)
torch.cuda.synchronize()
full = m.capture(*fr, branches=1)
print("one batch-8 graph        : %.1f frames/s" % timeit(lambda: full(*fr)))
a = full(*fr).clone()
for sizes in ([4, 4], [5, 3], [6, 2], [3, 3, 2], [4, 2, 2], [4, 4]):
    g, outs = branches(sizes)
    fps = timeit(g.replay)
    g.replay(); torch.cuda.synchronize()
    same = torch.equal(a, torch.cat(outs, 0))
    print("concurrent branches %-10s: %.1f frames/s  same bits: %s" % (sizes, fps, same))

# ---- skewed branches: branch B starts when branch A's encoder is done (A decoder || B encoder) ----
def enc(model, image, sparse, valid, k):
    x = torch.cat([sparse, valid], dim=1)
    d = model.sparse_to_dense_pool(x)
    latent, skips = model.encoder(image, d, k)
    return latent, skips, d.shape[-2:]

def dec(model, latent, skips, shape):
    feats = model.decoder.features(latent, skips, shape)
    return kb.ops.depth_head(feats, model.decoder.output0.conv.weight, model.min_predict_depth, model.max_predict_depth)

parts = [[f[i * 4:(i + 1) * 4].contiguous() for f in fr] for i in range(2)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur)
    with torch.cuda.stream(sA):
        eA = enc(m, *parts[0])
        mid = sA.record_event()
        oA = dec(m, *eA)
    sB.wait_event(mid)
    with torch.cuda.stream(sB):
        oB = dec(m, *enc(m, *parts[1]))
    cur.wait_stream(sA); cur.wait_stream(sB)
print("skewed 2 x 4 (B starts after A's encoder): %.1f frames/s  same bits: %s" % (timeit(g.replay), torch.equal(a, torch.cat([oA, oB], 0))))

# ---- small skews: branch B starts after branch A's S2D / after A's whole encoder level 0 ----
def staged(model, image, sparse, valid, k, mark=None, stream=None):
    x = torch.cat([sparse, valid], dim=1)
    d = model.sparse_to_dense_pool(x)
    ev = stream.record_event() if (mark == "s2d" and stream is not None) else None
    latent, skips = model.encoder(image, d, k)
    feats = model.decoder.features(latent, skips, d.shape[-2:])
    out = kb.ops.depth_head(feats, model.decoder.output0.conv.weight, model.min_predict_depth, model.max_predict_depth)
    return out, ev

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur)
    with torch.cuda.stream(sA):
        oA, ev = staged(m, *parts[0], mark="s2d", stream=sA)
    sB.wait_stream(cur)
    sB.wait_event(ev)
    with torch.cuda.stream(sB):
        oB, _ = staged(m, *parts[1])
    cur.wait_stream(sA); cur.wait_stream(sB)
print("2 x 4, B starts after A's S2D            : %.1f frames/s  same bits: %s" % (timeit(g.replay), torch.equal(a, torch.cat([oA, oB], 0))))

# ---- free-running half-batch streams: no join per step, optional half-period phase offset --------------
# Each half batch has its own encoder graph and decoder graph; stream A and stream B replay theirs back to back.
# With `offset`, B starts when A's first encoder is done, so that A's decoder (MFMA-bound) overlaps B's encoder
# (HBM-heavy) from then on -- unlike the in-graph skew above there is no tail per step, only one at the very end.
def capture_pair(part, stream):
    for _ in range(2):
        dec(m, *enc(m, *part))
    torch.cuda.synchronize()
    ge, gd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(ge, stream=stream):
        e = enc(m, *part)
    with torch.cuda.graph(gd, stream=stream):
        o = dec(m, *e)
    return ge, gd, o

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
geA, gdA, oA = capture_pair(parts[0], sA)
geB, gdB, oB = capture_pair(parts[1], sB)
torch.cuda.synchronize()

def free_run(reps, offset):
    torch.cuda.synchronize()
    t = time.perf_counter()
    ev = None
    with torch.cuda.stream(sA):
        geA.replay()
        ev = sA.record_event()
        gdA.replay()
    if offset:
        sB.wait_event(ev)
    with torch.cuda.stream(sB):
        geB.replay(); gdB.replay()
    for _ in range(reps - 1):
        with torch.cuda.stream(sA):
            geA.replay(); gdA.replay()
        with torch.cuda.stream(sB):
            geB.replay(); gdB.replay()
    torch.cuda.synchronize()
    return 8 * reps / (time.perf_counter() - t)

for offset in (False, True):
    free_run(3, offset)
    for reps in (20, 60):
        print("free-running 2 x 4 streams, offset=%s, %d steps: %.1f frames/s  same bits: %s" %
              (offset, reps, free_run(reps, offset), torch.equal(a, torch.cat([oA, oB], 0))))
