#!/usr/bin/env python3
"""How much of a graph-replay step is host/copy overhead: replay with caller tensors (copied into the static
buffers every step) vs replay with the static buffers themselves vs two graphs alternating on two streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
dev = torch.device("cuda:0")
cfg = kb.kitti_config()
m = kb.modules.KBNetModel.from_config(cfg, dev)
m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN["kitti"]))
fr = [f.to(dev) for f in kb.synthetic.make_frames(8, 352, 1216, "kitti", seed=1)]
def timeit(fn, reps=40):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return 8 * reps / (time.perf_counter() - t)
g1 = m.capture(*fr)
print("replay(caller tensors)  : %.1f frames/s" % timeit(lambda: g1(*fr)))
print("replay(static buffers)  : %.1f frames/s" % timeit(lambda: g1(*g1.static_in)))
print("graph.replay() only     : %.1f frames/s" % timeit(g1.graph.replay))
g2 = m.capture(*fr)
s = [torch.cuda.Stream(), torch.cuda.Stream()]
gs = [g1, g2]
k = [0]
def alt():
    i = k[0] & 1; k[0] += 1
    with torch.cuda.stream(s[i]):
        gs[i].graph.replay()
print("two graphs, two streams : %.1f frames/s" % timeit(alt))
