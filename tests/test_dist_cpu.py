"""CPU, world_size 2, gloo: frame sharding + output all-gather plumbing (dist.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import kbnet_amd as kb


def test_shard_bounds_partition():
    for n in (1, 7, 8, 256):
        for world in (1, 2, 3, 8):
            spans = [kb.dist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_forward(image, sparse, valid, k):
    # a frame-independent stand-in for the HIP forward: enough to check routing/order
    return image.mean(1, keepdim=True) + sparse + valid * k[:, 0, 0].view(-1, 1, 1, 1)


def _worker(rank, world, port, n_total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, _, w = kb.dist.init("gloo")
    assert (r, w) == (rank, world)
    frames = kb.synthetic.make_frames(n_total, 16, 24, "kitti", seed=5)
    local = kb.dist.shard_frames(frames, rank, world)
    runner = kb.dist.ShardedRunner(_fake_forward, rank, world)
    for _ in range(2):  # second call re-uses the gather buffer
        out = runner.step(local, n_total=n_total)
    ref = _fake_forward(*frames)
    ok = torch.equal(out, ref)
    if n_total % world == 0:  # pipelined mode: result of step i arrives with step i+1 / drain()
        prunner = kb.dist.ShardedRunner(_fake_forward, rank, world)
        assert prunner.step_pipelined(local) is None
        ok = ok and torch.equal(prunner.step_pipelined(local), ref)
        ok = ok and torch.equal(prunner.step_pipelined(local), ref)
        ok = ok and torch.equal(prunner.drain(), ref)
    else:  # ragged shards cannot go through the fixed-size pipelined gather: refused, not hung
        prunner = kb.dist.ShardedRunner(_fake_forward, rank, world)
        try:
            prunner.step_pipelined(local)
            ok = False
        except ValueError:
            pass
    t = kb.dist.max_over_ranks(float(rank), torch.device("cpu"))
    kb.dist.barrier()
    q.put((rank, ok, t))
    dist.destroy_process_group()


# BASELINE config 5: a stream mixing VOID 480x640, NYUv2 416x576 and KITTI 352x1216 frames, per-frame
# intrinsics, sharded over the ranks with one gather per shape bucket.
MIXED = [("void", 480, 640, 3), ("void", 416, 576, 1), ("kitti", 352, 1216, 4)]   # (preset, H, W, frames in the step)


def _mixed_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    kb.dist.init("gloo")
    runner = kb.dist.ShardedRunner(_fake_forward, rank, world)
    ok = True
    for step in range(2):   # second step re-uses the per-shape buffers
        buckets, refs = [], []
        for i, (preset, h, w, n_total) in enumerate(MIXED):
            frames = list(kb.synthetic.make_frames(n_total, h, w, preset, seed=10 * step + i))
            frames[3] = frames[3] * (1.0 + 0.1 * torch.arange(n_total).view(-1, 1, 1) / n_total)  # per-frame K
            lo, hi = kb.dist.shard_bounds(n_total, rank, world)
            local = [t[lo:hi] for t in frames] if hi > lo else None   # 1 frame over 2 ranks: rank 1 holds none
            fwd = (lambda s: (lambda *a: _fake_forward(*a) * s))(float(i + 1))   # "its own weights" per bucket
            buckets.append((fwd, local, n_total, (1, h, w)))
            refs.append(_fake_forward(*frames) * float(i + 1))
        outs = runner.step_mixed(buckets)
        ok = ok and len(outs) == len(refs) and all(torch.equal(o, r) for o, r in zip(outs, refs))
    kb.dist.barrier()
    q.put((rank, ok, 1.0))
    dist.destroy_process_group()


def _spawn(target, world, *args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=90) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def test_mixed_shape_stream_world2_gloo():
    results = _spawn(_mixed_worker, 2)
    assert all(ok for _, ok, _ in results)


def test_pipelined_world1_returns_detached_previous_step():
    """world 1: the 'previous step' result must not alias the forward's (re-used) output buffer."""
    static = torch.zeros(2, 1, 4, 4)

    def fwd(a, b, c, d):
        static.copy_(a)
        return static            # like a captured graph's static output

    r = kb.dist.ShardedRunner(fwd, 0, 1)
    x1, x2 = torch.ones(2, 1, 4, 4), torch.full((2, 1, 4, 4), 2.0)
    assert r.step_pipelined((x1, None, None, None)) is None
    prev = r.step_pipelined((x2, None, None, None))
    assert torch.equal(prev, x1)
    assert torch.equal(r.drain(), x2)


@pytest.mark.parametrize("n_total", [4, 5])
def test_sharded_runner_world2_gloo(n_total):
    results = _spawn(_worker, 2, n_total)
    assert all(ok for _, ok, _ in results)
    assert all(t == 1.0 for _, _, t in results)
