"""GPU tests of the drop-in boundary: the two north-star modules patched into a
reference-shaped model the way INTEGRATION.md section 2 shows, non-default streams, several devices in one
process, a checkpoint written by the reference's own `save_model`, and the RCCL path of dist.py.

    python -m pytest tests -m gpu -q
"""
import os
import socket
import types

import pytest
import torch

import kbnet_amd as kb
from conftest import GOLDEN_DIR, load_golden
from oracle import kbnet_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north_star: "within 1e-4 relative fp32"


@pytest.mark.gpu
def test_concurrent_host_threads_share_one_model(dev):
    """SURVEY 8(b) threading: the library keeps no mutable state between calls and launches on the caller's stream, ctypes drops the GIL
    around every call -- DataParallel-style host threads may drive ONE model at once.  Four threads, each on a stream of its own with
    inputs of its own, 12 forwards each, against the same forwards issued one after the other."""
    import threading
    cfg = kb.kitti_config().narrow()
    m = kb.modules.KBNetModel.from_config(cfg, dev)
    m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=1, gain=1.3))
    inputs = [[f.to(dev) for f in kb.synthetic.make_frames(2, 64, 96 + 32 * (t % 2), "kitti", seed=20 + t)] for t in range(4)]
    expect = [m.forward(*fr).clone() for fr in inputs]      # also packs every weight blob once, before the threads start
    torch.cuda.synchronize()
    results, errors = [None] * 4, []

    def work(t):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                outs = [m.forward(*inputs[t]) for _ in range(12)]
            st.synchronize()
            results[t] = outs
        except Exception as e:   # noqa: BLE001 (reported by the main thread)
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(4):
        assert all(torch.equal(o, expect[t]) for o in results[t]), t


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("GPU tests need a visible MI355X (run with -m gpu on a GPU box)")
    kb._lib.load()
    return torch.device("cuda:0")


# ----------------------------------------------------------------------------------------------------------
# A stand-in with the SHAPE of the reference's model code (the reference itself cannot travel to the GPU box):
# `networks` / `net_utils` namespaces whose two north-star symbols are replaced exactly as INTEGRATION.md
# section 2 patches the reference's modules, an encoder that builds dense coordinates with torch ops like
# reference src/networks.py:317-352 (incl. the level-1-ratio quirk) and calls the KB blocks BY KEYWORD
# (:368-372, :401-405), and a model forward that calls S2D POSITIONALLY (src/kbnet_model.py:166-184).
# Every other layer of the stand-in is a plain torch.nn.functional op, i.e. what the unpatched reference runs.
# ----------------------------------------------------------------------------------------------------------
def _patched_namespaces():
    networks, net_utils = types.SimpleNamespace(), types.SimpleNamespace()
    networks.SparseToDensePool = kb.modules.SparseToDensePool
    net_utils.CalibratedBackprojectionBlock = kb.modules.CalibratedBackprojectionBlock
    return networks, net_utils


def _torch_conv(x, w, stride=1, slope=0.2):
    y = torch.nn.functional.conv2d(x, w, stride=stride, padding=w.shape[-1] // 2)
    return y if slope is None else torch.nn.functional.leaky_relu(y, slope)


class _ReferenceShapedModel:
    def __init__(self, cfg, sds, device):
        networks, net_utils = _patched_namespaces()
        act = torch.nn.LeakyReLU(negative_slope=0.20, inplace=True)
        fi, fd = cfg.n_filters_encoder_image, cfg.n_filters_encoder_depth
        self.cfg = cfg
        self.sd_enc = {k: v.to(device) for k, v in sds[1].items()}
        self.sd_dec = {k: v.to(device) for k, v in sds[2].items()}
        self.sparse_to_dense_pool = networks.SparseToDensePool(
            input_channels=cfg.input_channels_depth, min_pool_sizes=list(cfg.min_pool_sizes_sparse_to_dense_pool),
            max_pool_sizes=list(cfg.max_pool_sizes_sparse_to_dense_pool),
            n_convolution=cfg.n_convolution_sparse_to_dense_pool, n_filter=cfg.n_filter_sparse_to_dense_pool,
            weight_initializer=cfg.weight_initializer, activation_func=cfg.activation_func)
        self.sparse_to_dense_pool.load_state_dict(sds[0])
        self.sparse_to_dense_pool = torch.nn.DataParallel(self.sparse_to_dense_pool, device_ids=[device.index]).to(device)
        self.kb = []
        for n in range(4):
            cf = fi[n - 1] + fi[n - 1] if n > 0 else fi[0]
            blk = net_utils.CalibratedBackprojectionBlock(
                in_channels_image=fi[max(n - 1, 0)], in_channels_depth=fd[max(n - 1, 0)], in_channels_fused=cf,
                n_filter_image=fi[n], n_filter_depth=fd[n], n_filter_fused=fi[n], n_convolution_image=1,
                n_convolution_depth=1, n_convolution_fused=1, weight_initializer=cfg.weight_initializer,
                activation_func=act)
            pref = f"calibrated_backprojection{n + 1}."
            blk.load_state_dict({k[len(pref):]: v for k, v in sds[1].items() if k.startswith(pref)})
            self.kb.append(blk.to(device))

    @staticmethod
    def _coordinates(k, n, h, w):       # reference src/networks.py:317-331 + src/net_utils.py:1601-1636
        xs = torch.linspace(0, w - 1, w, device=k.device)
        ys = torch.linspace(0, h - 1, h, device=k.device)
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        xy_h = torch.stack([xx, yy, torch.ones_like(xx)], 0).unsqueeze(0).repeat(n, 1, 1, 1)
        return torch.matmul(torch.inverse(k), xy_h.view(n, 3, -1)).view(n, 3, h, w)

    def forward(self, image, sparse_depth, validity_map_depth, intrinsics):
        input_depth = torch.cat([sparse_depth, validity_map_depth], dim=1)
        input_depth = self.sparse_to_dense_pool(input_depth)                      # positional, through DataParallel
        n, _, h0, w0 = image.shape
        e = self.sd_enc
        conv_image = _torch_conv(image, e["conv0_image.conv.weight"])
        conv_depth = _torch_conv(input_depth, e["conv0_depth.conv.weight"])
        h1, w1 = (h0 + 1) // 2, (w0 + 1) // 2
        scale = torch.tensor([[w1 / w0, 1.0, w1 / w0], [1.0, h1 / h0, h1 / h0], [1.0, 1.0, 1.0]], device=image.device)
        fused, skips, h, w = None, [], h0, w0
        for lvl in range(4):
            k = intrinsics if lvl == 0 else intrinsics * scale               # Q1: level-1 ratio at every deeper level
            coordinates = self._coordinates(k, n, h, w)
            conv_image, conv_depth, fused = self.kb[lvl](image=conv_image, depth=conv_depth, coordinates=coordinates,
                                                         fused=fused)       # by keyword, dense coordinates
            skips.append(torch.cat([fused, conv_depth], 1))
            h, w = (h + 1) // 2, (w + 1) // 2
        c5i = _torch_conv(fused, e["conv5_image.conv_block.0.conv.weight"], 2)
        c5d = _torch_conv(conv_depth, e["conv5_depth.conv_block.0.conv.weight"], 2)
        x = torch.cat([c5i, c5d], 1)
        d = self.sd_dec
        for name, skip in (("deconv4", skips[3]), ("deconv3", skips[2]), ("deconv2", skips[1]), ("deconv1", skips[0]),
                           ("deconv0", None)):
            size = skip.shape[2:4] if skip is not None else (h0, w0)
            x = _torch_conv(torch.nn.functional.interpolate(x, size=size, mode="nearest"),
                            d[f"{name}.deconv.conv.conv.weight"])
            x = _torch_conv(torch.cat([x, skip], 1) if skip is not None else x, d[f"{name}.conv.conv.weight"])
        logits = _torch_conv(x, d["output0.conv.weight"], slope=None)
        dmin, dmax = self.cfg.min_predict_depth, self.cfg.max_predict_depth
        return dmin / (torch.sigmoid(logits) + dmin / dmax)


def test_patched_reference_shaped_model_on_side_stream(dev):
    """INTEGRATION.md section 2 end to end: the two HIP modules inside otherwise stock torch code (torch's own
    convs for the rest, as in the unpatched reference), positional S2D call through DataParallel, keyword KB
    calls with DENSE coordinates, everything on a non-default stream.  Checked against the oracle."""
    cfg = kb.kitti_config().narrow()
    sds = kb.synthetic.make_state_dicts(cfg, seed=4, gain=1.3)
    frames = kb.synthetic.make_frames(2, 96, 160, "kitti", seed=3, jitter_intrinsics=0.1)
    ref = orc.kbnet_forward(*frames, *sds, cfg.min_pools, cfg.max_pools, cfg.min_predict_depth,
                            cfg.max_predict_depth)
    model = _ReferenceShapedModel(cfg, sds, dev)
    dframes = [f.to(dev) for f in frames]
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side), torch.no_grad():
        out = model.forward(*dframes)
        # nothing ran on the default stream: the modules launch on torch's CURRENT stream
    side.synchronize()
    err = float(((out.cpu() - ref).abs() / ref.abs()).max())
    assert err < TOL, err
    # the all-HIP model on the same stream gives the same answer within the same bar
    full = kb.modules.KBNetModel.from_config(cfg, dev)
    full.load_state_dicts(*sds)
    with torch.cuda.stream(side):
        out2 = full.forward(*dframes)
    side.synchronize()
    assert float(((out2.cpu() - ref).abs() / ref.abs()).max()) < TOL


def test_configs1_hybrid_full_size_batch8(dev):
    """BASELINE configs[1] as it is written: KITTI 352x1216, batch 8, fp32, one MI355X, "S2D + KB-layer HIP kernels, convs still
    PyTorch-ROCm" -- the two north-star modules patched into the reference-shaped model at FULL width and
    FULL size, every other layer a torch.nn.functional op (MIOpen).  The first and the last frame go through the oracle; the all-HIP
    model on the same frames is reported beside it."""
    cfg = kb.kitti_config()
    sds = kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN["kitti"])
    frames = kb.synthetic.make_frames(8, 352, 1216, "kitti", seed=1, jitter_intrinsics=0.1)
    model = _ReferenceShapedModel(cfg, sds, dev)
    dframes = [f.to(dev) for f in frames]
    with torch.no_grad():
        out = model.forward(*dframes)
    torch.cuda.synchronize(dev)
    assert tuple(out.shape) == (8, 1, 352, 1216)
    full = kb.modules.KBNetModel.from_config(cfg, dev)
    full.load_state_dicts(*sds)
    out_hip = full.forward(*dframes)
    worst = worst_hip = 0.0
    for i in (0, 7):
        ref = orc.kbnet_forward(*[f[i:i + 1] for f in frames], *sds, cfg.min_pools, cfg.max_pools, cfg.min_predict_depth,
                                cfg.max_predict_depth)
        worst = max(worst, float(((out[i:i + 1].cpu() - ref).abs() / ref.abs()).max()))
        worst_hip = max(worst_hip, float(((out_hip[i:i + 1].cpu() - ref).abs() / ref.abs()).max()))
    print(f"configs[1] hybrid (HIP S2D + KB blocks, torch convs) 8x352x1216: max rel err vs oracle {worst:.3e}; all-HIP {worst_hip:.3e}")
    assert worst < TOL, worst
    assert worst_hip < TOL, worst_hip


def test_two_devices_in_one_process():
    """Kernel attributes (dynamic LDS limit) are per device and the launches follow the tensors' device, not the
    current one: a model on cuda:1 while cuda:0 is current (the reference accepts any `device`; DataParallel
    replicas run on per-device threads, src/kbnet_model.py:408-415)."""
    if not torch.cuda.is_available():
        pytest.skip("needs GPUs")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices")
    cfg = kb.kitti_config().narrow()
    sds = kb.synthetic.make_state_dicts(cfg, seed=0, gain=1.3)
    frames = kb.synthetic.make_frames(2, 64, 96, "kitti", seed=1)
    outs = []
    torch.cuda.set_device(0)
    for index in (0, 1, 0):
        d = torch.device("cuda", index)
        m = kb.modules.KBNetModel.from_config(cfg, d)
        m.load_state_dicts(*sds)
        outs.append(m.forward(*[f.to(d) for f in frames]).cpu())   # current device stays 0 throughout
        assert torch.cuda.current_device() == 0
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    dp = torch.nn.DataParallel(kb.modules.SparseToDensePool(2, [5, 7], [9], 8, 3, "xavier_normal", "leaky_relu"),
                               device_ids=[0, 1]).to("cuda:0")
    x = torch.cat(kb.synthetic.make_frames(4, 32, 64, "kitti", seed=2)[1:3], 1).to("cuda:0")
    whole = dp.module(x)
    assert torch.equal(dp(x), whole), "DataParallel scatter over two devices must reproduce the single-device result"


def test_reference_written_checkpoint_restores_and_runs(dev):
    """f3: tests/golden/ckpt_kitti_narrow.pth was written by the REFERENCE's KBNetModel.save_model
    (src/kbnet_model.py:353-376; gen_golden.py gen_checkpoint) with the weights of golden fwd_kitti.
    restore_model (the counterpart of :378-406) -> forward must reproduce the reference's own output."""
    g = load_golden("fwd_kitti")
    m = kb.modules.KBNetModel.from_config(kb.kitti_config().narrow(), dev)
    step, _ = m.restore_model(os.path.join(GOLDEN_DIR, "ckpt_kitti_narrow.pth"))
    assert step == 1234
    out = m.forward(*[g[k].to(dev) for k in ("image", "sparse_depth", "validity_map", "intrinsics")])
    err = float(((out.cpu() - g["output_depth"]).abs() / g["output_depth"].abs()).max())
    assert err < TOL, err


# ------------------------------------------------------------------------------------------------ RCCL
def _rccl_single_rank_body():
    """(runs in a child process: see test_rccl_single_rank_group)  One-GPU boxes cannot host two RCCL ranks, but a ONE-rank group is legal: the whole RCCL path of dist.py --
    group creation, the async all-gather on RCCL's stream ordered against HIP-graph replays, the pipelined ring of
    buffers, ragged step(), the shape buckets of step_mixed -- runs on hardware (ShardedRunner(gather_single=True))."""
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        cfg = kb.kitti_config().narrow()
        m = kb.modules.KBNetModel.from_config(cfg, dev)
        m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=1.3))
        a = [t.to(dev) for t in kb.synthetic.make_frames(4, 64, 96, "kitti", seed=1, jitter_intrinsics=0.1)]
        b = [t.to(dev) for t in kb.synthetic.make_frames(4, 64, 96, "kitti", seed=2, jitter_intrinsics=0.1)]
        ea, eb = m.forward(*a).clone(), m.forward(*b).clone()
        replay = m.capture(*a)
        runner = kb.dist.ShardedRunner(replay, 0, 1, gather_single=True)
        assert runner.collective
        assert runner.step_pipelined(a) is None
        assert torch.equal(runner.step_pipelined(b), ea)      # result of step i arrives with step i+1
        assert torch.equal(runner.step_pipelined(a), eb)
        assert torch.equal(runner.drain(), ea)
        eager = kb.dist.ShardedRunner(m.forward, 0, 1, gather_single=True)
        assert torch.equal(eager.step(b, n_total=4), eb)
        c = [t.to(dev) for t in kb.synthetic.make_frames(2, 32, 64, "kitti", seed=3)]
        ec = m.forward(*c).clone()
        outs = eager.step_mixed([(m.forward, a, 4, (1, 64, 96)), (m.forward, c, 2, (1, 32, 64))])
        assert torch.equal(outs[0], ea) and torch.equal(outs[1], ec)
        assert kb.dist.max_over_ranks(3.0, dev) == 3.0
        kb.dist.barrier()
        # the form bench.py runs: two rotating graph outputs, the all-gather reads the graph's own output tensor (no staging copy),
        # and bench.timed_steps around it -- graph replay + async gather + drain() ordering with the real collective
        replay2 = m.capture(*a, outputs=2)
        r2 = kb.dist.ShardedRunner(replay2, 0, 1, gather_single=True)
        assert r2.step_pipelined(a) is None
        assert torch.equal(r2.step_pipelined(b), ea)
        assert torch.equal(r2.step_pipelined(a), eb)
        assert torch.equal(r2.step_pipelined(b), ea)
        assert torch.equal(r2.drain(), eb)
        assert all(slot[0] is None or slot[0].data_ptr() in [t.data_ptr() for t in replay2.static_outs] for slot in r2._ring)
        import bench
        r3 = kb.dist.ShardedRunner(m.capture(*a, outputs=2), 0, 1, gather_single=True)
        elapsed, gathered = bench.timed_steps(r3, a, steps=6, warmup=3, dev=dev)
        assert elapsed > 0 and torch.equal(gathered, ea)
        # captures while the process group's watchdog thread polls the events of fresh collectives: with the default (global) capture
        # error mode one full-suite run in four died here or in one of the captures above -- "operation not permitted when stream is
        # capturing" thrown on the watchdog thread (profiles/r06/v99_rccl_child_failure.err); GraphedForward captures thread-locally
        t = torch.ones(1 << 20, device=dev)
        for _ in range(24):
            for _ in range(4):
                dist.all_reduce(t, async_op=True)
            g = m.capture(*a)
            assert torch.equal(g(*a), ea)
        torch.cuda.synchronize()
        print("RCCL_BODY_OK", flush=True)
    finally:
        dist.destroy_process_group()


def test_rccl_single_rank_group(dev):
    """The one-rank RCCL group test above, in a child process: every assertion of the body must pass (the child prints RCCL_BODY_OK
    after the last one).  The teardown is outside the claim: ProcessGroupNCCL's destroy aborted the interpreter once in this round's
    runs (profiles/r06/README.md, v97) after a body that had passed -- inside the suite's own process that abort would have ended
    the whole `-m gpu` run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_boundary_gpu as t; t._rccl_single_rank_body()"
                        % (root, os.path.join(root, "tests"))], capture_output=True, text=True, env=env, timeout=600)
    if "RCCL_BODY_OK" not in r.stdout:   # the whole stderr of the child, where the run's artefacts are kept
        dump = os.path.join(root, "gpurun_out")
        if os.path.isdir(dump):
            with open(os.path.join(dump, "rccl_child_failure.err"), "w") as f:
                f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
        print(r.stderr[-6000:])
    assert "RCCL_BODY_OK" in r.stdout, "the child died before the end of the body (stderr printed above)"
    if r.returncode != 0:
        print(f"(the child's teardown exited with {r.returncode} after the body had passed: {r.stderr[-300:]})")




def test_bench_two_ranks_on_one_gpu_over_gloo(dev):
    """bench.py's WHOLE N > 1 path on a one-GPU box ("the N > 1 path has never executed on hardware"): the driver's
    own command line, `python bench.py --gpus 2 ...`, spawns its two ranks; KBN_BENCH_TEST_BACKEND=gloo-cuda makes both drive cuda:0 and
    sends the collectives through gloo (RCCL refuses two ranks on one device).  Real model, real graph capture with two rotating outputs,
    pipelined in-place gather, barrier + max-over-ranks timing, the sustained run, rank 0's roofline pass, ranks leaving together -- only
    RCCL itself is not in it (a ONE-rank RCCL group runs in test_rccl_single_rank_group).  The printed rate is two ranks time-sharing
    one GPU, not a scaling point."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KBN_BENCH_TEST_BACKEND="gloo-cuda", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--frames-per-gpu", "4"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line, the other rank none"
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["n_ranks_seen"] == 2 and c["frames_per_gpu"] == 4 and c["global_batch"] == 8
    assert c["gathered_frames"] == 8 and c["gather_matches_local_forward_rank0"] is True
    assert d["value"] > 0 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert c["sustained"]["steps"] >= 300 and d["roofline"]["frac"] > 0
    assert "gloo" in c["collective_backend"]
    _check_multi_gpu_fields(d, 2)


def _check_multi_gpu_fields(d, world):
    """The fields that make an N-rank line explain itself: per-rank rates, the collective alone, rank 0 alone, the
    efficiency against it, and configs[4]'s mixed-shape stream under the same ranks."""
    c, mg = d["config"], d["config"]["multi_gpu"]
    assert c["n_ranks_seen"] == world == d["n_gpus"]
    assert len(mg["per_rank_frames_per_s"]) == world and min(mg["per_rank_frames_per_s"]) > 0
    assert len(mg["forward_only"]["per_rank_frames_per_s"]) == world
    assert mg["allgather_ms"] > 0 and mg["allgather"]["bytes_gathered_per_rank"] == world * c["frames_per_gpu"] * 352 * 1216 * 4
    assert mg["rank0_alone_frames_per_s"] > 0 and 0 < mg["scaling_efficiency"] < 1.5
    assert c["mixed_shape_stream_frames_per_s"] > 0, "configs[4] in miniature runs under the N ranks"
    assert d["pipe"] == "fp16x3-split"


def test_bench_all_visible_gpus_over_rccl():
    """The driver's command line on every visible device with the REAL backend: `python bench.py --gpus N` (N = device count >= 2), RCCL
    all-gather over xGMI.  One-GPU leases skip it -- with the reason in the report (-rs), never silently; the gloo-cuda test above runs the
    same code path there."""
    import json
    import subprocess
    import sys
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n < 2:
        pytest.skip(f"{n} visible device(s): an N > 1 RCCL run needs two (SCALE_rNN is the driver's 8-GPU run of this command)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "KBN_BENCH_TEST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "10", "--warmup", "3"],
                       capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["collective_backend"] == "nccl" and d["config"]["gather_matches_local_forward_rank0"] is True
    _check_multi_gpu_fields(d, n)
    print(json.dumps(d["config"]["multi_gpu"]))


def _rccl_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    r, local, w = kb.dist.init("nccl")
    dev = torch.device("cuda", local)
    cfg = kb.kitti_config().narrow()
    m = kb.modules.KBNetModel.from_config(cfg, dev)
    m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=1.3))
    per = 4
    frames = kb.synthetic.make_frames(per * world, 64, 96, "kitti", seed=1, jitter_intrinsics=0.1)
    local_frames = [t.to(dev) for t in kb.dist.shard_frames(frames, rank, world)]
    replay = m.capture(*local_frames)
    runner = kb.dist.ShardedRunner(replay, rank, world)
    assert runner.step_pipelined(replay.static_in) is None
    first = runner.step_pipelined(replay.static_in).clone()
    last = runner.drain().clone()
    single = m.forward(*[t.to(dev) for t in frames])      # all frames on one rank, eager
    ok = torch.equal(first, single) and torch.equal(last, single)
    # ragged step(): 2 * world - 1 frames
    n_total = per * world - 1
    sub = [t[:n_total] for t in frames]
    got = kb.dist.ShardedRunner(m.forward, rank, world).step(
        [t.to(dev) for t in kb.dist.shard_frames(sub, rank, world)], n_total=n_total)
    ok = ok and torch.equal(got, single[:n_total])
    kb.dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sharded_runner_rccl():
    """dist.ShardedRunner over RCCL (backend 'nccl'): the real graphed forward on every rank, pipelined all-gather,
    bitwise equal to one rank computing all frames.  Needs two devices (two RCCL ranks cannot share one GPU)."""
    if not torch.cuda.is_available():
        pytest.skip("needs GPUs")
    world = min(2, torch.cuda.device_count())
    if world < 2:
        pytest.skip("needs two visible devices for two RCCL ranks")
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok in results)
