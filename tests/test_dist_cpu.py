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
    t = kb.dist.max_over_ranks(float(rank), torch.device("cpu"))
    kb.dist.barrier()
    q.put((rank, ok, t))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [4, 5])
def test_sharded_runner_world2_gloo(n_total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    assert all(t == 1.0 for _, _, t in results)
