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


class _RotatingForward:
    """What modules.GraphedForward(outputs=2) looks like to the runner: two static output tensors, calls alternate."""
    rotating_outputs = 2

    def __init__(self):
        self.outs = [torch.zeros(2, 1, 4, 4), torch.zeros(2, 1, 4, 4)]
        self.turn = 0

    def next_output(self):
        return self.outs[self.turn]

    def __call__(self, a, b, c, d):
        out = self.outs[self.turn]
        self.turn ^= 1
        out.copy_(a)
        return out


def test_pipelined_world1_rotating_outputs_are_not_copied():
    """A forward with two rotating outputs is gathered IN PLACE: the ring holds the forward's own tensors (no staging copy) and the
    previous step's result survives the current step's forward."""
    fwd = _RotatingForward()
    r = kb.dist.ShardedRunner(fwd, 0, 1)
    xs = [torch.full((2, 1, 4, 4), float(i)) for i in range(1, 5)]
    assert r.step_pipelined((xs[0], None, None, None)) is None
    for i in (1, 2, 3):
        prev = r.step_pipelined((xs[i], None, None, None))
        assert torch.equal(prev, xs[i - 1])
        assert prev.data_ptr() in [o.data_ptr() for o in fwd.outs], "the ring slot IS the forward's output tensor"
    assert torch.equal(r.drain(), xs[3])
    # a call from outside shifts the rotation against the runner's step count: results stay right (the runner goes by tensor identity)
    fwd(xs[0], None, None, None)
    assert r.step_pipelined((xs[1], None, None, None)) is not None
    assert torch.equal(r.step_pipelined((xs[2], None, None, None)), xs[1])
    assert torch.equal(r.drain(), xs[2])


def _rotating_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    kb.dist.init("gloo")
    fwd = _RotatingForward()
    runner = kb.dist.ShardedRunner(fwd, rank, world)
    mine = lambda step: torch.full((2, 1, 4, 4), float(10 * step + rank))
    whole = lambda step: torch.cat([torch.full((2, 1, 4, 4), float(10 * step + r)) for r in range(world)])
    ok = runner.step_pipelined((mine(0), None, None, None)) is None
    for step in range(1, 5):
        prev = runner.step_pipelined((mine(step), None, None, None))
        ok = ok and torch.equal(prev, whole(step - 1))
    ok = ok and torch.equal(runner.drain(), whole(4))
    ok = ok and all(slot[0].data_ptr() in [o.data_ptr() for o in fwd.outs] for slot in runner._ring)
    fwd(mine(9), None, None, None)     # an outside call shifts the rotation: the next steps must still gather the right tensors
    for step in range(5, 8):
        prev = runner.step_pipelined((mine(step), None, None, None))
        ok = ok and (step == 5 or torch.equal(prev, whole(step - 1)))
    ok = ok and torch.equal(runner.drain(), whole(7))
    kb.dist.barrier()
    q.put((rank, bool(ok), 1.0))
    dist.destroy_process_group()


def test_pipelined_rotating_outputs_world2_gloo():
    results = _spawn(_rotating_worker, 2)
    assert all(ok for _, ok, _ in results)


# ---- BASELINE configs[3]'s rank arithmetic: 256 frames over EIGHT ranks (gloo, tiny frames, stand-in forward) --------
MIXED8 = [("void", 24, 32, 19), ("void", 20, 28, 5), ("kitti", 16, 40, 256)]   # 19 and 5 frames over 8 ranks: ragged, empty ranks


def _world8_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    kb.dist.init("gloo")
    n_total = 256
    frames = kb.synthetic.make_frames(n_total, 8, 12, "kitti", seed=7)
    lo, hi = kb.dist.shard_bounds(n_total, rank, world)
    ok = (hi - lo) == 32 and lo == 32 * rank                  # configs[3]: 32 frames per GPU, contiguous
    local = kb.dist.shard_frames(frames, rank, world)
    ref = _fake_forward(*frames)
    runner = kb.dist.ShardedRunner(_fake_forward, rank, world)
    ok = ok and torch.equal(runner.step(local, n_total=n_total), ref)
    prunner = kb.dist.ShardedRunner(_fake_forward, rank, world)
    ok = ok and prunner.step_pipelined(local) is None
    ok = ok and torch.equal(prunner.step_pipelined(local), ref)
    ok = ok and torch.equal(prunner.drain(), ref)
    # a ragged global batch (250 frames: ranks 0-1 hold 32, the others 31) goes through the padded slots of step()
    sub = [t[:250] for t in frames]
    ok = ok and torch.equal(kb.dist.ShardedRunner(_fake_forward, rank, world).step(kb.dist.shard_frames(sub, rank, world), n_total=250), ref[:250])
    # mixed-shape buckets: ranks without a frame of a shape still join that bucket's gather
    buckets, refs = [], []
    for i, (preset, h, w, n) in enumerate(MIXED8):
        fr = list(kb.synthetic.make_frames(n, h, w, preset, seed=20 + i))
        a, b = kb.dist.shard_bounds(n, rank, world)
        buckets.append((_fake_forward, [t[a:b] for t in fr] if b > a else None, n, (1, h, w)))
        refs.append(_fake_forward(*fr))
    outs = runner.step_mixed(buckets)
    ok = ok and all(torch.equal(o, r) for o, r in zip(outs, refs))
    t = kb.dist.max_over_ranks(float(rank), torch.device("cpu"))
    per_frame = torch.full((hi - lo, 4), float(rank))
    mean = kb.dist.mean_metrics_over_ranks(per_frame)
    ok = ok and bool(torch.allclose(mean, torch.full((4,), 3.5, dtype=torch.float64)))
    kb.dist.barrier()
    q.put((rank, bool(ok), t))
    dist.destroy_process_group()


def test_sharded_runner_world8_gloo_256_frames():
    """BASELINE configs[3] (KITTI batch 256 over 8 GPUs) without the GPUs: shard_bounds, step, step_pipelined, a ragged batch, the
    mixed-shape buckets of configs[4] with ranks that hold no frame of a shape, max / mean reductions -- at WORLD SIZE 8."""
    results = _spawn(_world8_worker, 8)
    assert len(results) == 8 and all(ok for _, ok, _ in results)
    assert all(t == 7.0 for _, _, t in results)


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
        mg = line["config"]["multi_gpu"]      # the self-explaining N-rank fields
        ok = ok and len(mg["per_rank_frames_per_s"]) == world and len(mg["forward_only"]["per_rank_frames_per_s"]) == world
        ok = ok and mg["allgather_ms"] > 0 and mg["rank0_alone_frames_per_s"] > 0 and mg["scaling_efficiency"] > 0
        ok = ok and line["pipe"] == "fp16x3-split" and line["dtype"] == "f32"
    else:
        ok = ok and not lines                  # only rank 0 prints
    q.put((rank, ok, seen["first"]))


def test_bench_rank_logic_world2_gloo(tmp_path):
    results = _spawn(_bench_worker, 2, str(tmp_path))
    assert all(ok for _, ok, _ in results)
    firsts = {r: f for r, _, f in results}
    assert firsts[0] != firsts[1], "ranks draw different frames (seed 1 + rank)"


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_refuses_gpus_that_contradict_world(monkeypatch):
    """Inside a torchrun environment `--gpus N` must equal WORLD_SIZE (a rank never re-spawns)."""
    bench = _load_bench()
    monkeypatch.setenv("WORLD_SIZE", "3")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    with pytest.raises(SystemExit, match="WORLD_SIZE=3"):
        bench.main(["--gpus", "2"], backend="gloo", forward_factory=lambda *a: _fake_forward)


def test_bench_plain_python_launch_spawns_its_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT torchrun's environment -- the form the driver uses --
    must not die on WORLD_SIZE: it re-launches itself as 2 ranks (bench.self_spawn), rank 0 prints ONE JSON line with
    n_gpus 2 and the exit code is the ranks'.  KBN_BENCH_TEST_BACKEND=gloo swaps the HIP forward for a stand-in on CPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["KBN_BENCH_TEST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--frames-per-gpu", "4"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["n_ranks_seen"] == 2 and line["config"]["gathered_frames"] == 8
    assert line["cpu_baseline"] is None
    mg = line["config"]["multi_gpu"]
    assert len(mg["per_rank_frames_per_s"]) == 2 and mg["allgather_ms"] > 0 and 0 < mg["scaling_efficiency"]
    # failing ranks' exit code comes back (here: zero timed steps cannot be averaged)
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "0", "--warmup", "0",
                          "--frames-per-gpu", "2"], env=env, capture_output=True, text=True, timeout=240)
    assert bad.returncode != 0


class _FakeEvent:
    def __init__(self, t):
        self.t = t

    def elapsed_time(self, other):
        return other.t - self.t   # ms


def test_bench_roofline_fractions_never_exceed_one():
    """every launch group is priced at the peak of the pipe ITS MFMAs run on (the record carries it),
    so no fraction of the bench line can exceed 1 -- round 3 priced kb1_front / kb1_depth_front / conv_tail /
    conv_split_1x1s2 (fp16 MFMAs) at the fp32 peak and reported whole_forward_frac 1.24."""
    bench = _load_bench()
    ev = lambda ms: (_FakeEvent(0.0), _FakeEvent(ms))
    steps = 2
    prof = []
    for _ in range(steps):
        # (name, work, executed, pipe, nbytes, start, end): fp16 kernels issue 3 x work (+ padding)
        prof.append(("conv_split", 347e9, 1056e9, "fp16", 1.15e9) + ev(0.96))
        prof.append(("kb1_front", 245e9, 850e9, "fp16", 1.5e9) + ev(1.04))
        prof.append(("kb1_depth_front", 60e9, 180e9, "fp16", 0.7e9) + ev(0.53))
        prof.append(("conv_tail", 35e9, 125e9, "fp16", 0.9e9) + ev(0.47))
        prof.append(("conv_split_1x1s2", 20e9, 70e9, "fp16", 0.3e9) + ev(0.18))
        prof.append(("conv_split_upfold", 300e9, 400e9, "fp16", 0.5e9) + ev(0.40))   # folded: executes < 3 x work
        prof.append(("conv_wino", 100e9, 44.4e9, "fp32", 0.4e9) + ev(0.40))            # Winograd: executes 4/9 of work
        prof.append(("conv_dma<3,2,4,2,4>", 5e9, 6e9, "fp32", 0.2e9) + ev(0.17))
        prof.append(("s2d", 0.55e9, 24.8e9, "fp32", 0.55e9) + ev(0.58))
        prof.append(("kb_xyz", 4e7, None, None, 4e7) + ev(0.012))
    groups = bench.summarise_profile(prof, steps)
    table = bench.per_kernel_table(groups, steps)
    assert set(table) == {r[0] for r in prof}
    for name, row in table.items():
        for key in ("useful_frac", "issued_frac", "hbm_frac"):
            assert row[key] is None or 0.0 < row[key] <= 1.0, (name, key, row[key])
        assert row["useful_frac"] is None or row["useful_frac"] <= row["issued_frac"]
    assert table["kb1_front"]["pipe"] == "fp16" and abs(table["kb1_front"]["issued_frac"] - 850e9 / 1.04e-3 / 2.5e15) < 1e-3
    assert table["kb_xyz"]["issued_frac"] is None and table["kb_xyz"]["hbm_frac"] is not None
    step_seconds = sum(s.elapsed_time(e) for *_, s, e in prof) * 1e-3 / steps
    whole = bench.pipe_seconds(groups) / steps / step_seconds
    assert 0.0 < whole <= 1.0
    with pytest.raises(ValueError):   # one name on two pipes would be priced wrongly: refused
        bench.summarise_profile(prof + [("conv_split", 1e9, 3e9, "fp32", 1e6) + ev(0.1)], steps)


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
