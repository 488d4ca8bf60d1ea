#!/usr/bin/env python3
"""BASELINE.json config 5 / SURVEY 8(d): an interleaved stream of VOID 480x640 and NYUv2 416x576 frames (VOID
preset) and KITTI 352x1216 frames (KITTI preset) with per-frame intrinsics (+-10 %), two weight sets resident,
one captured graph per (shape, local batch).

    python tools/mixed_stream_bench.py [frames_per_shape_per_step]                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        tools/mixed_stream_bench.py [frames_per_shape_per_step]                          # N GPUs, RCCL

A step holds B frames of each shape; every shape bucket is split over the ranks with dist.shard_bounds and gathered
by its own all-gather (dist.ShardedRunner.step_mixed: per-shape buffers, the gather of bucket i in flight while
bucket i+1 computes).  Prints frames/s of the interleaved stream next to the per-shape rates (rank 0)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, kbnet_amd as kb

rank, local_rank, world = kb.dist.init("nccl")
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8 * world
models = {}
for preset in ("kitti", "void"):
    cfg = kb.PRESETS[preset]()
    m = kb.modules.KBNetModel.from_config(cfg, dev)
    m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN[preset]))
    models[preset] = m
stream = [("void", "void", (480, 640)), ("void", "nyu_v2", (416, 576)), ("kitti", "kitti", (352, 1216))]
local, replay = {}, {}
for preset, stats, shape in stream:
    frames = kb.synthetic.make_frames(B, *shape, stats, seed=3, jitter_intrinsics=0.1)   # the step's frames of this shape
    lo, hi = kb.dist.shard_bounds(B, rank, world)
    if hi > lo:
        local[shape] = [f[lo:hi].to(dev) for f in frames]
        replay[shape] = models[preset].capture(*local[shape])
runner = kb.dist.ShardedRunner(None, rank, world)


def buckets(order):
    return [(replay.get(shape), replay[shape].static_in if shape in replay else None, B, (1,) + shape) for _, _, shape in order]


def rate(order, reps):
    runner.step_mixed(buckets(order))
    torch.cuda.synchronize()
    kb.dist.barrier()
    t = time.perf_counter()
    for _ in range(reps):
        outs = runner.step_mixed(buckets(order))
    torch.cuda.synchronize()
    kb.dist.barrier()
    dt = kb.dist.max_over_ranks(time.perf_counter() - t, dev)
    assert all(o.shape[0] == B for o in outs)
    return reps * len(order) * B / dt


for s in stream:
    r = rate([s], 10)
    if rank == 0:
        print(f"{s[1]:7s} {s[2][0]}x{s[2][1]} alone : {r:8.1f} frames/s", flush=True)
r = rate(stream, 10)
if rank == 0:
    print(f"interleaved stream ({B} frames per shape and step over {world} rank(s), VOID/NYU/KITTI round robin): {r:8.1f} frames/s")
if world > 1:
    torch.distributed.destroy_process_group()
