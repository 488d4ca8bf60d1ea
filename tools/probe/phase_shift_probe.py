#!/usr/bin/env python3
"""Do the two sub-batch branches of a step run faster when they are OUT of phase?

Inside the shipped graph both branches start together every replay: their issue-bound front kernels overlap with each other and their
power-bound concat convs overlap with each other.  Here each sub-batch is a graph of its own (GraphedForward split_graphs) and the two
streams run free, K replays each, the second one delayed by a fraction of a step.  Also: two whole-batch graphs (one branch each) in
flight on two streams.  frames/s over K steps, everything drained at the end."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb
dev = torch.device("cuda:0")
N, K = 32, int(os.environ.get("K", "60"))
cfg = kb.kitti_config()
m = kb.modules.KBNetModel.from_config(cfg, dev)
m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN["kitti"]))
fr = [f.to(dev) for f in kb.synthetic.make_frames(N, 352, 1216, "kitti", seed=1)]

def rate(fn, steps=K, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return N * steps / (time.perf_counter() - t)

# how long is _sleep(1e6)?
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(10000000); e1.record(); torch.cuda.synchronize()
ms_per_mcycle = e0.elapsed_time(e1) / 10.0
print(f"_sleep: {ms_per_mcycle:.4f} ms per 1e6 cycles", flush=True)

gF = m.capture(*fr)
print("fused graph, 2 branches (shipped)      : %.1f" % rate(lambda: gF.graph.replay()), flush=True)
gS = m.capture(*fr, split_graphs=True)
ref = gF(*fr).clone()
assert torch.equal(gS(*fr), ref)
print("one graph per branch, joined per step   : %.1f" % rate(lambda: gS(*gS.static_in)), flush=True)
gA, gB = gS.graphs[0]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def free_run(delay_ms, steps=K):
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.cuda.stream(sB):
        if delay_ms > 0:
            torch.cuda._sleep(int(delay_ms / ms_per_mcycle * 1e6))
    for _ in range(steps):
        with torch.cuda.stream(sA): gA.replay()
        with torch.cuda.stream(sB): gB.replay()
    torch.cuda.synchronize()
    return N * steps / (time.perf_counter() - t - delay_ms * 1e-3 * 0.0)

free_run(0, 5)
for d in (0.0, 1.5, 3.0, 4.5, 6.0, 8.0):
    print(f"free-running branches, B delayed {d:4.1f} ms: {free_run(d):.1f}   (delay inside the timed region)", flush=True)
assert torch.equal(gS.static_outs[0], ref)

# two whole-batch single-branch graphs in flight
g1 = m.capture(*fr, branches=1)
g2 = m.capture(*fr, branches=1)
def free_run2(delay_ms, steps=K):
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.cuda.stream(sB):
        if delay_ms > 0:
            torch.cuda._sleep(int(delay_ms / ms_per_mcycle * 1e6))
    for i in range(steps):
        if i & 1:
            with torch.cuda.stream(sB): g2.graph.replay()
        else:
            with torch.cuda.stream(sA): g1.graph.replay()
    torch.cuda.synchronize()
    return N * steps / (time.perf_counter() - t)
free_run2(0, 4)
for d in (0.0, 3.0, 6.0, 9.0):
    print(f"two whole-batch graphs in flight, second delayed {d:4.1f} ms: {free_run2(d):.1f}", flush=True)
print("fused graph again                       : %.1f" % rate(lambda: gF.graph.replay()), flush=True)
