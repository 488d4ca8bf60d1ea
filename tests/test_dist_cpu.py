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


# ---- bench.py's rank logic, end to end at world size 2 (gloo, stand-in forward) --------------------------------------
# What the first real `bench.py --gpus 8` executes besides the HIP forward: the --gpus / WORLD_SIZE check, per-rank frame
# seeds, the timed region (barrier + synchronise on both sides, exactly K steps, MAX over ranks), the pipelined
# all-gather, rank-0-only printing of ONE JSON line, cpu_baseline null for N > 1.
def _bench_worker(rank, world, port, outdir, q):
    import contextlib
    import importlib.util
    import io
    import json
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench.HEIGHT, bench.WIDTH = 16, 24          # tiny frames: this is a plumbing test
    seen = {}

    def factory(r, dev, frames):
        seen["rank"], seen["dev"], seen["first"] = r, dev, float(frames[0].flatten()[0])
        return _fake_forward

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--frames-per-gpu", "4"],
                         backend="gloo", forward_factory=factory)
    lines = [ln for ln in buf.getvalue().splitlines() if ln.strip()]
    ok = seen["rank"] == rank and seen["dev"].type == "cpu"
    if rank == 0:
        ok = ok and len(lines) == 1
        line = json.loads(lines[0])
        ok = ok and line["n_gpus"] == world and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
        ok = ok and line["cpu_baseline"] is None and line["vs_baseline"] is None and line["higher_is_better"] is True
        ok = ok and line["config"]["global_batch"] == 4 * world and line["config"]["gathered_frames"] == 4 * world
        ok = ok and abs(line["value"] - 4 * world * 3 / (line["ms_per_step"] * 3e-3)) < 1e-2 * line["value"]
    else:
        ok = ok and not lines                  # only rank 0 prints
    q.put((rank, ok, seen["first"]))


def test_bench_rank_logic_world2_gloo(tmp_path):
    results = _spawn(_bench_worker, 2, str(tmp_path))
    assert all(ok for _, ok, _ in results)
    firsts = {r: f for r, _, f in results}
    assert firsts[0] != firsts[1], "ranks draw different frames (seed 1 + rank)"


def test_bench_refuses_gpus_without_matching_world(monkeypatch):
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit, match="WORLD_SIZE=1"):
        bench.main(["--gpus", "2"], backend="gloo", forward_factory=lambda *a: _fake_forward)


# ---- evaluation metrics over a sharded run (reference src/kbnet.py:932-984 averages over ALL samples) ----------------
def _metrics_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    kb.dist.init("gloo")
    g = torch.Generator().manual_seed(3)
    per_frame = torch.rand(5, 4, generator=g, dtype=torch.float64) * 100.0     # what ops.eval_metrics returns per frame
    lo, hi = kb.dist.shard_bounds(5, rank, world)                               # ragged: 3 + 2 frames
    got = kb.dist.mean_metrics_over_ranks(per_frame[lo:hi])
    ok = torch.allclose(got, per_frame.mean(dim=0), rtol=1e-12, atol=0)
    empty = kb.dist.mean_metrics_over_ranks(per_frame[:0] if rank == 1 else per_frame)   # a rank without frames
    ok = ok and torch.allclose(empty, per_frame.mean(dim=0), rtol=1e-12, atol=0)
    kb.dist.barrier()
    q.put((rank, ok, 1.0))
    dist.destroy_process_group()


def test_eval_metrics_mean_over_ranks_world2_gloo():
    assert all(ok for _, ok, _ in _spawn(_metrics_worker, 2))
    one = kb.dist.mean_metrics_over_ranks(torch.tensor([[1.0, 2.0, 3.0, 4.0], [3.0, 4.0, 5.0, 6.0]]))   # no process group
    assert torch.equal(one, torch.tensor([2.0, 3.0, 4.0, 5.0], dtype=torch.float64))
