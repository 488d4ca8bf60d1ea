#!/usr/bin/env python3
"""Benchmark of the KBNet inference hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms work for any N: started WITHOUT torchrun's environment (no WORLD_SIZE) and with --gpus N > 1, the script
re-launches itself as N ranks under `torch.distributed.run --standalone`-style arguments on 127.0.0.1 (`self_spawn`),
passes the ranks' output through and exits with their exit code.

A step = one `KBNetModel.forward` over this rank's batch of synthetic KITTI-shaped
frames (352 x 1216, fp32, inputs resident in HBM), followed -- for N > 1 -- by the RCCL
all-gather of the depth maps.  Frames shard across ranks (weak scaling: 32 frames per GPU =
BASELINE.json configs[3]'s per-GPU share of its 256 frames and configs[2]'s single-GPU batch in
fp32; the batch-8 rate of configs[1] is reported as a side field); there is no other collective on
the data path.  Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around
every launch of the dominant kernel (the MFMA conv variant with the largest total time):
`achieved` / `frac` are the ALGORITHMIC FLOPs of its launches (the reference's direct-conv
formulation, 2 * N * Hout * Wout * Cin * 9 * Cout) over their time and over the dense peak of the pipe
its MFMAs run on; a split-operand kernel takes three fp16 products per fp32 product, so its ceiling is
1/3 (`ceiling_frac`); `issued` carries the FLOPs of the MFMA instructions it really issues (padding that
is issued included: the figure SQ_INSTS_MFMA x 32768 of the PMC collection gives);
`config.sustained` repeats the timed region for >= 300 steps and >= 3 s (and replaces `value` when it is
more than 3 % lower), `config.batch1` is the graph-replayed latency of one frame;
`cpu_baseline` times the CPU oracle (the port of the reference, bit-identical to it) on
this box's host cores over a bounded sample.
"""

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import kbnet_amd as kb  # noqa: E402

HEIGHT, WIDTH = 352, 1216
FRAMES_PER_GPU = 32      # configs[2] (fp32 leg) / configs[3] per-GPU share; --frames-per-gpu 8 = configs[1]
SIDE_BATCH = 8           # configs[1]
FP32_MFMA_PEAK_TFLOPS = kb.ops.PIPE_PEAK_TFLOPS["fp32"]  # MI355X_MICROARCH.md: v_mfma_f32_*_f32, dense (the fp32 vector datapath)
FP16_MFMA_PEAK_TFLOPS = kb.ops.PIPE_PEAK_TFLOPS["fp16"]  # MI355X_MICROARCH.md: dense bf16/fp16 MFMA (the matrix core proper)
# host-level launch name -> the rocprofv3 kernel-name prefix of its instantiations (PMC lookup)
SPLIT_KERNELS = {"conv_split": "conv3x3_split_kernel<0, 8,", "conv_split_up": "conv3x3_split_kernel<1, 8,",
                 "conv_split_s2": "conv3x3_split_kernel<2, 2,", "conv_split_upfold": "upconv2x_split_kernel"}
HBM_PEAK_GBS = 8000.0
# switches that take the forward off the parity-gated path (throughput-only arithmetic) or off the shipped kernels
NON_PARITY_KNOBS = ("KBN_FP16_ONE_TERM", "KBN_NO_SPLIT")
README_KITTI_MS_PER_FRAME = 15.19   # reference README.md:232 (batch 1, its GPU, preprocessing included, no device sync)


class knob_env:
    """`with knob_env("KBN_NO_SPLIT"):` sets a KBN_* switch to 1 for a side leg and puts the caller's value (or its absence) back."""

    def __init__(self, name, value="1"):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = os.environ.get(self.name)
        os.environ[self.name] = self.value
        kb.ops.reload_env()

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.old
        kb.ops.reload_env()
        return False


def replay_rate(replay, inputs, reps, warm, frames, world, dev):
    """frames/s of `reps` graph replays between barriers + synchronises (MAX over ranks)."""
    for _ in range(warm):
        replay(*inputs)
    torch.cuda.synchronize()
    kb.dist.barrier()
    t = time.perf_counter()
    for _ in range(reps):
        out = replay(*inputs)
    torch.cuda.synchronize()
    kb.dist.barrier()
    return frames * world * reps / kb.dist.max_over_ranks(time.perf_counter() - t, dev), out


def summarise_profile(prof, steps):
    """Per-kernel-group sums of an ops.PROFILE list: {name: dict(work, seconds, launches, executed, pipe, nbytes)}.
    The PIPE each launch's MFMAs run on travels with the record (ops._launch), so every group is priced at the peak of
    its own pipe -- the split-operand kernels (conv_split*, conv_split_1x1s2, kb1_front, kb1_depth_front, conv_tail) at
    the fp16 matrix core's 2.5 PFLOP/s, the fp32-MFMA kernels at 157.3 TFLOP/s."""
    groups = {}
    for name, work, executed, pipe, nbytes, s, e in prof:
        g = groups.setdefault(name, {"work": 0.0, "seconds": 0.0, "launches": 0, "executed": 0.0, "has_executed": True,
                                     "pipe": pipe, "nbytes": 0.0, "has_bytes": True})
        g["work"] += work
        g["seconds"] += s.elapsed_time(e) * 1e-3
        g["launches"] += 1
        if executed is None:
            g["has_executed"] = False
        else:
            g["executed"] += executed
        if nbytes is None:
            g["has_bytes"] = False
        else:
            g["nbytes"] += nbytes
        if pipe != g["pipe"]:
            raise ValueError(f"launch group {name} mixes pipes {g['pipe']} and {pipe}")
    return groups


def per_kernel_table(groups, steps):
    """roofline.per_kernel: for every launch group of a step its time and the fraction of each roof it reaches --
    `issued_frac` = FLOPs of the MFMAs it issues / time / peak of ITS pipe, `useful_frac` = the reference's
    multiply-adds (x the products its pipe needs per fp32 product) / time / the same peak, `hbm_frac` = algorithmic
    bytes (inputs once + outputs once) / time / 8 TB/s.  None where a roof does not apply.  No fraction can exceed 1
    unless a launch executes fewer FLOPs than the reference formulation (`useful_frac` of Winograd / folded up-convs on
    the fp32 pipe): those are capped by `issued_frac`, which counts what runs."""
    table = {}
    for name, g in groups.items():
        t = g["seconds"]
        row = {"us_per_step": round(t / steps * 1e6, 1), "launches_per_step": round(g["launches"] / steps, 2), "pipe": g["pipe"],
               "useful_frac": None, "issued_frac": None, "hbm_frac": None}
        if g["pipe"] is not None and g["has_executed"] and g["executed"] > 0 and t > 0:
            peak = kb.ops.PIPE_PEAK_TFLOPS[g["pipe"]] * 1e12
            row["issued_frac"] = round(g["executed"] / t / peak, 4)
            row["useful_frac"] = round(min(kb.ops.PIPE_PRODUCTS[g["pipe"]] * g["work"], g["executed"]) / t / peak, 4)
        if g["has_bytes"] and g["nbytes"] > 0 and t > 0:
            row["hbm_frac"] = round(g["nbytes"] / t / (HBM_PEAK_GBS * 1e9), 4)
        table[name] = row
    return table


def pipe_seconds(groups):
    """Seconds the issued MFMA FLOPs of every group would take at the dense peak of the pipe they run on."""
    return sum(g["executed"] / (kb.ops.PIPE_PEAK_TFLOPS[g["pipe"]] * 1e12) for g in groups.values()
               if g["pipe"] is not None and g["has_executed"] and g["executed"] > 0)
WEIGHT_GAIN = kb.synthetic.PARITY_GAIN["kitti"]  # logits std ~ 1: the sigmoid head is exercised over its range (synthetic.PARITY_GAIN)


def conv_gflop_per_frame(cfg, h, w):
    """Algorithmic conv work of one forward (2*MAC), from the parameter shapes and the
    resolution each conv runs at (100.722 GFLOP at 352x1216)."""
    sizes = [(h, w)]
    for _ in range(5):
        sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
    px = [a * b for a, b in sizes]
    total = 0.0
    for k, s in kb.config.s2d_param_shapes(cfg).items():
        total += 2.0 * px[0] * s[0] * s[1] * s[2] * s[3]
    for k, s in kb.config.encoder_param_shapes(cfg).items():
        macs = s[0] * s[1] * s[2] * s[3]
        if k.startswith("conv0_"):
            lvl = 0
        elif k.startswith("conv5_"):
            lvl = 5
        elif "proj_depth" in k:
            lvl = int(k.split(".")[0][-1]) - 1  # evaluated by the reference at full block resolution
        else:
            lvl = int(k.split(".")[0][-1]) if "calibrated" in k else int(k[4])
        total += 2.0 * px[lvl] * macs
    for k, s in kb.config.decoder_param_shapes(cfg).items():
        lvl = 0 if k.startswith("output0") else int(k[6])
        total += 2.0 * px[lvl] * s[0] * s[1] * s[2] * s[3]
    return total / 1e9


def lookup_traffic(kernel, launches, frames_per_gpu):
    """PMC evidence for `kernel` from the committed collection (tools/collect_pmc.py: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE / MFMA-busy in separate passes, FETCH_SIZE doubled on gfx950): (HBM bytes per launch, detail) or
    (None, None)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not files:
        return None, None
    doc = json.load(open(files[-1]))
    if doc.get("frames_per_gpu", 8) != frames_per_gpu:   # bytes per launch scale with the batch: only a matching collection counts
        return None, None
    table = doc["kernels"]
    if kernel == "conv_wino":
        want = "conv_wino_kernel<0>"
    elif kernel in SPLIT_KERNELS:
        want = SPLIT_KERNELS[kernel]
    else:
        m = re.match(r"(conv_\w+)<([\d,]+)>", kernel)
        if not m:
            return None, None
        want = m.group(1) + "_kernel<" + m.group(2).replace(",", ", ")
    # a host-level launch runs ONE kernel of the family -- which instantiation depends on the layer (with / without the
    # transposed tiles of a narrow last column, pair / fp32 sources): the launch-weighted mean over the instantiations
    hits = [e for name, e in table.items() if want in name and e.get("launches")]
    if not hits:
        return None, None
    total = sum(e["launches"] for e in hits)
    per = lambda key: sum(e.get(key, 0.0) * e["launches"] for e in hits) / total
    detail = {"source": os.path.relpath(files[-1], ROOT), "fetch_bytes": per("fetch_bytes"), "write_bytes": per("write_bytes"),
              "launches": total, "instantiations": len(hits)}
    if all("mfma_busy_frac" in e for e in hits):   # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs), time-weighted
        w = [e["launches"] * e.get("avg_us_under_pmc", 1.0) for e in hits]
        detail["mfma_busy_frac_pmc"] = round(sum(e["mfma_busy_frac"] * wi for e, wi in zip(hits, w)) / sum(w), 4)
    return per("hbm_bytes_per_launch"), detail


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(cfg, sds, frames, batch_frames=None, runs=3):
    """Oracle (CPU port of the reference, bit-identical to it) on this box's host cores, as SURVEY.md 8(d) asks: CPU
    model and host core count stated, N = 1 (one frame at a time like the reference's own loop, src/kbnet.py:887-921)
    and N = 8 (one batch-8 forward), MEDIAN of `runs` >= 3 runs each.  `value` is the better of the two rates (the
    reference's loader may batch).  Thread count: the reference's single-channel MaxPool2d / strided 1x1 conv do not
    scale past a few dozen threads (measured on the 256-core GPU box: 16 threads 0.98 s/frame, 64: 1.19, 128: 1.71,
    256: 33), so min(cores, KBNET_CPU_THREADS or 16) threads run and that number is reported as `cores` / `threads`.
    Returns (dict, oracle output of frame 0)."""
    from oracle import kbnet_oracle as orc
    host_cores = os.cpu_count() or 1
    threads = min(host_cores, int(os.environ.get("KBNET_CPU_THREADS", "16")))
    torch.set_num_threads(threads)
    run = lambda fr: orc.kbnet_forward(*fr, *sds, cfg.min_pools, cfg.max_pools, cfg.min_predict_depth, cfg.max_predict_depth)
    one = [f[0:1] for f in frames]
    ref = run(one)  # warm-up (also the parity reference for frame 0 of rank 0)

    def median_seconds(fr):
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            run(fr)
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), ts

    s1, t1 = median_seconds(one)
    res = {"value": 1.0 / s1, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
           "cpu_model": cpu_model_name(), "host_cores": host_cores, "threads": torch.get_num_threads(),
           "n1_frames_per_s": round(1.0 / s1, 4), "n1_seconds_per_run": [round(t, 3) for t in t1], "n8_frames_per_s": None,
           "statistic": f"median of {runs} runs",
           "sample": f"{runs} runs of 1 KITTI 352x1216 frame (batch 1)"}
    if batch_frames is not None and batch_frames[0].shape[0] > 1:
        nb = batch_frames[0].shape[0]
        sb, tb = median_seconds(batch_frames)
        res["n8_frames_per_s"] = round(nb / sb, 4)
        res["n8_seconds_per_run"] = [round(t, 3) for t in tb]
        res["value"] = max(res["value"], nb / sb)
        res["sample"] += f" + {runs} runs of one batch of {nb} frames"
    res["sample"] += ", fp32, oracle/kbnet_oracle.py (torch CPU)"
    return res, ref


def rccl_version():
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:   # no GPU build / no RCCL: the line still prints
        return None


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_spawn(argv, gpus: int, backend: str):
    """`python bench.py --gpus N` started without torchrun's environment: run the same command line as N ranks (one
    process per GPU) under torch.distributed.run on 127.0.0.1, pass their stdout / stderr through (rank 0 prints the ONE
    JSON line) and return their exit code.  Replaces the three DataParallel wrappers of reference
    src/kbnet_model.py:408-415 as the way more than one GPU is driven."""
    if backend in ("nccl", "gloo-cuda") and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if backend == "nccl":
        have = torch.cuda.device_count()
        if have < gpus:
            raise SystemExit(f"--gpus {gpus} but only {have} GPU(s) are visible")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    env["KBN_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def setup_ranks(gpus: int, backend: str):
    """One process per GPU, launched as the contract says: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
    environment (torch.distributed.run sets them; `main` re-launches itself under it when they are missing), `--gpus N`
    must equal WORLD_SIZE.  backend "nccl" = RCCL: the rank's device is cuda:LOCAL_RANK; "gloo": CPU
    (tests/test_dist_cpu.py runs this very code at world size 2)."""
    rank, local_rank, world = kb.dist.env_world()
    if world != gpus:
        raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus {gpus}` (it spawns its "
                         f"ranks) or `python -m torch.distributed.run --nproc-per-node {gpus} bench.py --gpus {gpus} ...`")
    if backend == "nccl":
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    elif backend == "gloo-cuda":
        # TEST MODE (KBN_BENCH_TEST_BACKEND=gloo-cuda): every rank drives cuda:0 and the collectives go through gloo, which moves CUDA
        # tensors between processes that share a device -- RCCL refuses two ranks on one GPU ("Duplicate GPU detected").  Lets a ONE-GPU
        # box run this script's whole N > 1 path (real graphed forward on every rank, pipelined gather, max-over-ranks timing, rank-0
        # roofline pass, ranks leaving together); the figure it prints is N ranks time-sharing one GPU, not a scaling point.
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
    else:
        dev = torch.device("cpu")
    kb.dist.init("gloo" if backend == "gloo-cuda" else backend)
    return rank, local_rank, world, dev


def mixed_stream_rate(dev, rank, world, frames_per_shape=8, reps=8):
    """BASELINE configs[4] in miniature on the ranks of this run: an interleaved stream of VOID 480x640 and NYUv2 416x576 frames (VOID
    preset) and KITTI 352x1216 frames (KITTI preset) with per-frame intrinsics (+-10 %), two weight sets resident, one captured graph per
    shape; every shape bucket is split over the ranks (dist.shard_bounds) and gathered by its own all-gather
    (ShardedRunner.step_mixed).  Returns frames/s of the round-robin stream (tools/mixed_stream_bench.py is the stand-alone form)."""
    b = frames_per_shape * world
    models, replay = {}, {}
    for preset in ("kitti", "void"):
        cfg = kb.PRESETS[preset]()
        m = kb.modules.KBNetModel.from_config(cfg, dev)
        m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN[preset]))
        models[preset] = m
    stream = [("void", "void", (480, 640)), ("void", "nyu_v2", (416, 576)), ("kitti", "kitti", (352, 1216))]
    for preset, stats, shape in stream:
        frames = kb.synthetic.make_frames(b, *shape, stats, seed=3, jitter_intrinsics=0.1)
        lo, hi = kb.dist.shard_bounds(b, rank, world)
        if hi > lo:
            replay[shape] = models[preset].capture(*[f[lo:hi].to(dev) for f in frames])
    runner = kb.dist.ShardedRunner(None, rank, world)
    buckets = [(replay.get(shape), replay[shape].static_in if shape in replay else None, b, (1,) + shape) for _, _, shape in stream]
    runner.step_mixed(buckets)
    torch.cuda.synchronize()
    kb.dist.barrier()
    t = time.perf_counter()
    for _ in range(reps):
        outs = runner.step_mixed(buckets)
    torch.cuda.synchronize()
    kb.dist.barrier()
    dt = kb.dist.max_over_ranks(time.perf_counter() - t, dev)
    assert all(o.shape[0] == b for o in outs)
    return reps * len(stream) * b / dt


def standin_forward(image, sparse, valid, k):
    """Frame-independent stand-in for the HIP forward (plumbing tests: KBN_BENCH_TEST_BACKEND=gloo)."""
    return image.mean(1, keepdim=True) + sparse + valid * k[:, 0, 0].view(-1, 1, 1, 1)


def device_sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)


def timed_steps(runner, step_inputs, steps: int, warmup: int, dev, per_rank: bool = False):
    """The contract's timed region: W untimed steps, then EXACTLY K steps bracketed by barrier + device synchronise on
    both sides; returns (seconds = MAX over ranks, gathered output of the last step).  A step = forward of this rank's
    frames + the asynchronous all-gather of the depth maps (the gather of step i overlaps step i+1's forward; the last
    one is drained inside the timed region)."""
    for _ in range(warmup):
        runner.step_pipelined(step_inputs)
    runner.drain()
    device_sync(dev)
    kb.dist.barrier()
    device_sync(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.step_pipelined(step_inputs)
    out = runner.drain()
    device_sync(dev)
    local = time.perf_counter() - t0      # this rank's own K steps (its last gather drained), before it waits for the others
    kb.dist.barrier()
    elapsed = time.perf_counter() - t0
    if per_rank:   # + every rank's own seconds, rank order (`config.multi_gpu.per_rank_frames_per_s`)
        return kb.dist.max_over_ranks(elapsed, dev), out, kb.dist.gather_over_ranks(local, dev)
    return kb.dist.max_over_ranks(elapsed, dev), out


def multi_gpu_report(forward, step_inputs, gathered_like, per, world, rank, dev, fps, local_seconds, steps, reps=20):
    """`config.multi_gpu`: what explains an N-rank figure without a second run.
      per_rank_frames_per_s   every rank's own rate over the timed steps (its clock stops when its last gather has drained)
      forward_only            all ranks replay their graph at once, no collective: per-rank rates, min / max
      allgather_ms            the step's collective alone: a loop of nothing but all_gather_into_tensor of the N x 1 x H x W maps
      rank0_alone             rank 0 replays while the other ranks wait at a barrier: the N = 1 rate of THIS job's build / box
      scaling_efficiency      value / (N x rank0_alone) -- the driver computes its own from the per-N runs; this one needs no history
    Every rank runs this (the loops hold collectives)."""
    import torch.distributed as tdist
    rep = {"per_rank_frames_per_s": [round(per * steps / t, 1) for t in local_seconds]}
    rep["per_rank_min_max"] = [min(rep["per_rank_frames_per_s"]), max(rep["per_rank_frames_per_s"])]

    def forward_loop():
        for _ in range(3):
            forward(*step_inputs)
        device_sync(dev)
        t = time.perf_counter()
        for _ in range(reps):
            forward(*step_inputs)
        device_sync(dev)
        return time.perf_counter() - t

    kb.dist.barrier()
    rates = [round(per * reps / t, 1) for t in kb.dist.gather_over_ranks(forward_loop(), dev)]
    rep["forward_only"] = {"per_rank_frames_per_s": rates, "min": min(rates), "max": max(rates), "sum": round(sum(rates), 1),
                           "what": f"{reps} graph replays per rank, all ranks at once, no collective"}
    if world > 1 and tdist.is_initialized():
        src = forward(*step_inputs)
        dst = torch.empty_like(gathered_like)
        for _ in range(3):
            tdist.all_gather_into_tensor(dst, src)
        device_sync(dev)
        kb.dist.barrier()
        t = time.perf_counter()
        for _ in range(reps):
            tdist.all_gather_into_tensor(dst, src)
        device_sync(dev)
        ag = kb.dist.max_over_ranks(time.perf_counter() - t, dev) / reps
        nbytes = dst.numel() * dst.element_size()
        rep["allgather_ms"] = round(1e3 * ag, 4)
        rep["allgather"] = {"bytes_gathered_per_rank": nbytes, "algbw_GBps": round(nbytes / ag / 1e9, 2),
                            "what": f"{reps} blocking all_gather_into_tensor calls of {per} x 1 x {HEIGHT} x {WIDTH} fp32 per rank, nothing else running; "
                                    "in the timed step the same collective runs under the next step's forward"}
        del dst
    else:
        rep["allgather_ms"] = None
    kb.dist.barrier()
    alone = forward_loop() if rank == 0 else None
    kb.dist.barrier()
    alone = kb.dist.gather_over_ranks(alone or 0.0, dev)[0]
    rep["rank0_alone_frames_per_s"] = round(per * reps / alone, 1)
    rep["scaling_efficiency"] = round(fps / (world * per * reps / alone), 4)
    rep["scaling_efficiency_what"] = "value / (n_gpus x rank0_alone_frames_per_s), same process, same build, same minute"
    return rep


def base_result(fps, world, steps, warmup, ms_per_step):
    """The keys the driver's contract names, in one place."""
    return {"metric": "depth-completion frames/sec at 352x1216", "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            # the pipe behind "f32": tensors / accumulators / results fp32; the wide 3x3 convs form each fp32 product from three fp16 MFMAs
            "pipe": "fp16x3-split", "data": "synthetic"}


def emit(result, rank: int):
    """Rank 0 prints the ONE JSON line."""
    if rank == 0:
        print(json.dumps(result), flush=True)


def main(argv=None, backend: str = "nccl", forward_factory=None):
    """`backend` / `forward_factory` are test hooks (tests/test_dist_cpu.py): with gloo and a stand-in forward --
    forward_factory(rank, dev, frames) -> callable -- the rank logic above runs end to end on CPU processes; the GPU-only
    measurements (roofline, side figures, cpu_baseline) are skipped then.  The same hook from the command line (so that
    the self-spawn path can be tested on CPU): KBN_BENCH_TEST_BACKEND=gloo runs `standin_forward` on tiny frames."""
    global HEIGHT, WIDTH
    argv = list(sys.argv[1:] if argv is None else argv)
    if os.environ.get("KBN_BENCH_TEST_BACKEND") == "gloo-cuda" and forward_factory is None:
        backend = "gloo-cuda"      # the REAL forward on every rank, all ranks on cuda:0, gloo collectives (setup_ranks)
    elif os.environ.get("KBN_BENCH_TEST_BACKEND") and forward_factory is None:
        backend = os.environ["KBN_BENCH_TEST_BACKEND"]
        forward_factory = lambda rank, dev, frames: standin_forward
        HEIGHT, WIDTH = 16, 24
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames-per-gpu", type=int, default=FRAMES_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-void", action="store_true", help="skip the VOID 480x640 side measurement")
    ap.add_argument("--no-side-batch", action="store_true", help="skip the batch-8 (configs[1]) side measurement")
    ap.add_argument("--no-bf16", action="store_true", help="accepted and ignored (the bf16 leg left the bench in round 5; the fp16 leg is configs[2]'s 16-bit figure)")
    ap.add_argument("--no-fp32-mfma", action="store_true", help="skip the all-fp32-MFMA side measurement (KBN_NO_SPLIT=1)")
    ap.add_argument("--branches", type=int, default=0,
                    help="concurrent sub-batches inside the captured graph (0 = default: 2 for even batches >= 4)")
    ap.add_argument("--eager", action="store_true", help="time plain launches instead of HIP-graph replay")
    ap.add_argument("--side", action="store_true",
                    help="with --gpus N > 1: also run the side measurements (VOID, batch 8, fp16 leg, fp32-MFMA-only, "
                         "unused conv); by default an N-rank run is the timed region plus rank 0's roofline pass")
    ap.add_argument("--no-fp16", action="store_true", help="skip the throughput-only one-term fp16 leg (configs[2])")
    ap.add_argument("--no-mixed", action="store_true", help="skip the mixed-shape stream side measurement (configs[4] in miniature)")
    ap.add_argument("--split-graphs", action="store_true",
                    help="one HIP graph per sub-batch branch, replayed on concurrent streams (GraphedForward(split_graphs=True)): lets the "
                         "encoder's level side branches run inside every sub-batch")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 300-step / >= 3 s sustained-rate run")
    ap.add_argument("--no-batch1", action="store_true", help="skip the batch-1 latency side measurement")
    ap.add_argument("--no-options", action="store_true", help="skip the side measurement of --deconv_type transpose / --activation_func relu | elu")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (never reached by a rank: torchrun sets WORLD_SIZE)
        if os.environ.get("KBN_BENCH_SPAWNED"):
            raise SystemExit("bench.py: spawned rank without WORLD_SIZE in its environment")
        raise SystemExit(self_spawn(argv, args.gpus, backend))

    rank, local_rank, world, dev = setup_ranks(args.gpus, backend)
    if forward_factory is None:
        # `value` is the parity-gated fp32 path: a caller's environment must not turn the timed region into one of the A/B or
        # throughput-only modes the side legs below switch on for themselves
        bad = [k for k in NON_PARITY_KNOBS if kb.ops.knob(k) != 0]
        if bad:
            raise SystemExit(f"bench.py: {', '.join(bad)} set in the environment -- the timed region would not be the parity-gated "
                             "path; unset it (the bench runs those modes itself as side legs)")
    if world > 1 and not args.side:   # (the mixed-shape stream stays: configs[4] is an N-rank workload)
        args.no_void = args.no_side_batch = args.no_bf16 = args.no_fp32_mfma = args.no_fp16 = args.no_batch1 = args.no_options = True
    per = args.frames_per_gpu
    # rank r holds frames [r*per, (r+1)*per) of the global batch (seed 1+rank; frame 0 of
    # rank 0 is the frame the CPU oracle sees)
    frames = kb.synthetic.make_frames(per, HEIGHT, WIDTH, "kitti", seed=1 + rank)
    frames = [f.to(dev) for f in frames]
    if forward_factory is not None:   # test hook: the plumbing alone
        runner = kb.dist.ShardedRunner(forward_factory(rank, dev, frames), rank, world)
        elapsed, out, local_seconds = timed_steps(runner, frames, args.steps, args.warmup, dev, per_rank=True)
        result = base_result(per * world * args.steps / elapsed, world, args.steps, args.warmup, 1e3 * elapsed / args.steps)
        multi_gpu = multi_gpu_report(runner.forward_fn, frames, out, per, world, rank, dev, result["value"], local_seconds, args.steps, reps=3)
        result["config"] = {"workload": "stand-in forward (plumbing test)", "frames_per_gpu": per, "global_batch": per * world,
                            "multi_gpu": multi_gpu,
                            "gathered_frames": int(out.shape[0]), "rank_seeds": [1 + r for r in range(world)],
                            "n_ranks_seen": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                            "backend": backend}
        result["roofline"] = None
        result["cpu_baseline"] = None
        emit(result, rank)
        if world > 1:
            torch.distributed.destroy_process_group()
        return result

    cfg = kb.kitti_config()
    sds = kb.synthetic.make_state_dicts(cfg, seed=0, gain=WEIGHT_GAIN)
    model = kb.modules.KBNetModel.from_config(cfg, dev)
    model.load_state_dicts(*sds)
    # The forward of this batch shape is captured once into a HIP graph; a step replays it (same
    # kernels, no per-launch host round trips).  --eager times the plain launch sequence instead.
    # --branches: concurrent sub-batches inside the graph (default: 2 for even batches >= 4, see GraphedForward)
    # outputs=2: the graph writes two alternating output tensors, so the asynchronous all-gather of step i reads the graph's own
    # output while step i+1's forward runs -- no staging copy inside the step (dist.ShardedRunner.step_pipelined)
    forward = model.forward if args.eager else model.capture(*frames, branches=args.branches or None, outputs=2,
                                                             split_graphs=args.split_graphs)
    runner = kb.dist.ShardedRunner(forward, rank, world)
    # Inputs live where the graph reads them (its static input tensors, filled once here): a producer such as
    # loader.InferenceFrameLoader writes there directly, so a step has no input copy.  Eager mode: the frames.
    step_inputs = forward.static_in if hasattr(forward, "static_in") else frames

    elapsed, out, local_seconds = timed_steps(runner, step_inputs, args.steps, args.warmup, dev, per_rank=True)
    out = out.clone()
    # this rank's slice of the gathered tensor must be the bits its own forward produces (the gather reads the graph's output in place)
    gather_ok = bool(torch.equal(out[rank * per:(rank + 1) * per], forward(*step_inputs)))
    if tuple(out.shape) != (per * world, 1, HEIGHT, WIDTH):
        raise SystemExit(f"gathered output has shape {tuple(out.shape)}, expected {(per * world, 1, HEIGHT, WIDTH)}")

    # Sustained rate: the SAME runner and graph for at least 300 more steps and at least 3 s (the part is power-capped: clocks settle
    # over seconds, and K = 50 steps is half a second).  Reported beside `value`; if it falls more than 3 % below, it REPLACES `value`.
    sustained = None
    if not args.eager and not args.no_sustained:
        ms = 1e3 * elapsed / args.steps
        n_sus = max(300, int(3000.0 / ms) + 1)
        sus_elapsed, _ = timed_steps(runner, step_inputs, n_sus, 0, dev)
        sustained = {"frames_per_s": round(per * world * n_sus / sus_elapsed, 1), "steps": n_sus, "seconds": round(sus_elapsed, 3)}

    # Per-kernel durations for the roofline: the same K steps launched eagerly, every ABI call
    # bracketed by HIP events on the launch stream (graph nodes cannot be timed individually).
    with kb.ops.autotune():      # whole-batch shapes: opt-in geometry tuning (the graph may run sub-batches)
        for _ in range(2):
            model.forward(*frames)
    torch.cuda.synchronize()
    kb.ops.PROFILE = []
    t1 = time.perf_counter()
    for _ in range(args.steps):
        model.forward(*frames)
    torch.cuda.synchronize()
    eager_ms = 1e3 * (time.perf_counter() - t1) / args.steps
    prof, kb.ops.PROFILE = kb.ops.PROFILE, None

    # "Reference-style" region (reference src/kbnet.py:896-921 times validity map + outlier removal +
    # image/255 + forward per sample): the same steps with the pre-model kernels in front.
    image255 = frames[0] * 255.0
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        img, valid, _ = kb.ops.preprocess(image255, frames[1])
        model.forward(img, frames[1], valid, frames[3])
    torch.cuda.synchronize()
    refstyle_ms = 1e3 * (time.perf_counter() - t2) / args.steps

    # Second shape north_star names (VOID 480 x 640, VOID preset, same batch): a short side measurement with
    # its own weights and graph; `value` above stays the KITTI metric.
    void_fps = None
    if not args.no_void:
        vcfg = kb.void_config()
        vmodel = kb.modules.KBNetModel.from_config(vcfg, dev)
        vmodel.load_state_dicts(*kb.synthetic.make_state_dicts(vcfg, seed=0, gain=kb.synthetic.PARITY_GAIN["void"]))
        vframes = [f.to(dev) for f in kb.synthetic.make_frames(per, 480, 640, "void", seed=1)]
        vreplay = vmodel.capture(*vframes)
        # all ranks run their frames at the same time: whole-job rate = frames of all ranks / slowest rank's time
        void_fps, _ = replay_rate(vreplay, vreplay.static_in, 10, 3, per, world, dev)
        del vreplay, vmodel, vframes

    # The reference's other architecture switches (run_kbnet.py --deconv_type transpose, --activation_func relu | elu) at the headline's
    # batch and size: what a user of those options gets -- the transposed decoder on the folded up-conv's split kernels, relu in the
    # shipped kernels (slope 0), elu layer by layer (activation passes of their own; the wide 3x3 convs still on the split-operand kernels).  Side measurements, forward only.
    options_fps = None
    if not args.no_options and not args.eager:
        import dataclasses
        options_fps = {}
        for label, changes in (("deconv_type=transpose", {"deconv_type": "transpose"}), ("activation_func=relu", {"activation_func": "relu"}),
                               ("activation_func=elu", {"activation_func": "elu"})):
            ocfg = dataclasses.replace(cfg, **changes)
            omodel = kb.modules.KBNetModel.from_config(ocfg, dev)
            omodel.load_state_dicts(*kb.synthetic.make_state_dicts(ocfg, seed=0, gain=kb.synthetic.PARITY_GAIN["kitti"]))
            oreplay = omodel.capture(*frames)
            fps, _ = replay_rate(oreplay, oreplay.static_in, 10, 3, per, world, dev)
            options_fps[label] = round(fps, 1)
            del oreplay, omodel

    # configs[1] (batch 8 per GPU, the round-1 headline) as a side measurement on the same weights
    side_fps = side_latency_fps = None
    if not args.no_side_batch and per != SIDE_BATCH:
        sframes = [f[:SIDE_BATCH].contiguous() for f in frames]
        sreplay = model.capture(*sframes)
        side_fps, _ = replay_rate(sreplay, sreplay.static_in, 20, 3, SIDE_BATCH, world, dev)
        del sreplay
        # the same batch in the latency form sized for its launches (two graph branches of four frames: set_latency_mode(frames=4))
        model.set_latency_mode(True, frames=SIDE_BATCH // 2)
        try:
            sreplay = model.capture(*sframes)
            side_latency_fps, _ = replay_rate(sreplay, sreplay.static_in, 20, 3, SIDE_BATCH, world, dev)
            del sreplay
        finally:
            model.set_latency_mode(False)

    # Batch-1 latency (graph replay of ONE frame, the encoder's level side branches on): the figure that sits beside the reference's
    # README.md:232 "15.19 ms per KITTI sample" (its GPU, batch 1, its timed region src/kbnet.py:896-921 incl. the pre-model stage)
    batch1 = None
    if not args.no_batch1 and not args.eager:
        one = [f[:1].contiguous() for f in frames]
        oreplay = model.capture(*one)
        rate1, _ = replay_rate(oreplay, oreplay.static_in, 200, 20, 1, 1, dev)
        image255_1 = one[0] * 255.0
        torch.cuda.synchronize()
        t9 = time.perf_counter()
        for _ in range(200):   # the reference's timed region: validity map + outlier removal + /255, then the forward
            img1, valid1, _ = kb.ops.preprocess(image255_1, one[1])
            oreplay(img1, one[1], valid1, one[3])
        torch.cuda.synchronize()
        ref_region_ms = 1e3 * (time.perf_counter() - t9) / 200
        # the latency form (KBNetModel.set_latency_mode: split-K on the launches one frame cannot fill the chip with; another summation
        # order than the default form, same 1e-4 bar): its own graph of the same frame
        model.set_latency_mode(True)
        try:
            lreplay = model.capture(*one)
            lrate1, lout = replay_rate(lreplay, lreplay.static_in, 200, 20, 1, 1, dev)
            lrel = float(((lout - oreplay(*one)).abs() / oreplay(*one).abs()).max())
            del lreplay
        finally:
            model.set_latency_mode(False)
        batch1 = {"ms_per_frame": round(1e3 / rate1, 4), "latency_mode_ms_per_frame": round(1e3 / lrate1, 4),
                  "latency_mode_max_rel_diff_vs_default_form": lrel,
                  "reference_style_region_ms_per_frame": round(ref_region_ms, 4),
                  "reference_readme_ms_per_frame": README_KITTI_MS_PER_FRAME,
                  "note": "graph replay of one KITTI 352x1216 frame; the reference's 15.19 ms (README.md:232) is on its own GPU and "
                          "includes the pre-model stage without a device sync: context, not a same-hardware comparison"}
        del oreplay

    # BASELINE configs[2] asks for a 16-bit figure: the one-term fp16 leg below is it.  (Rounds 2-4 also reported a bf16 leg -- bf16 MFMA
    # operands in the wide 3x3 convs over fp32 tensors, csrc/conv_bf16.hip / KBNetModel.set_bf16(): 3219 frames/s at mean error 3.8e-3.  It
    # predates the pair tensors, is slower AND less accurate than the fp16 leg (bf16 carries three mantissa bits fewer at the same MFMA
    # rate), and left the bench line in round 5; the kernels and their tests stay.)
    mine = out[rank * per:(rank + 1) * per]
    # The one-term leg (extended to every split-operand kernel in round 5): the SAME tuned split / pair kernels issuing
    # h1 w1 alone -- plain fp16 operands, fp32 accumulation, one MFMA instead of three, the h2 halves of the pair tensors neither written
    # nor fetched (KBN_FP16_ONE_TERM=1: concat convs, folded up-convs, stride-2 image convs, 1x1 stride-2 convs, both front kernels, the tail).  Two readings: (a) how MFMA-bound the three-product kernels are (same skeleton, a third of the MFMAs), (b) BASELINE
    # configs[2]'s 16-bit figure on the tuned kernels.  THROUGHPUT-ONLY: reported with its measured error, never `value`.
    fp16_leg = None
    if not args.no_fp16 and not args.eager:
        with knob_env("KBN_FP16_ONE_TERM"):
            hreplay = model.capture(*frames, branches=args.branches or None)
            hfps, hout = replay_rate(hreplay, hreplay.static_in, 10, 3, per, world, dev)
            hrel = (hout - mine).abs() / mine.abs()
            fp16_leg = {"frames_per_s": round(hfps, 1),
                        "scope": "EVERY split-operand kernel -- concat convs, 64- and 16-filter folded up-convs, stride-2 image convs, the 1x1 "
                                 "stride-2 conv_fused of KB3 / KB4, the encoder front (conv0 + KB1 + level 1's conv_fused, both branches) and the "
                                 "decoder tail: 93 % of the step -- with ONE fp16 MFMA per product (h1 w1: fp16 operands, fp32 accumulation), "
                                 "pair tensors read and written at 2 B / value; S2D and the conv_depth of KB2-4 (fp32 MFMA) as in `value`",
                        "max_rel_err_vs_fp32_path": float(hrel.max()), "mean_rel_err_vs_fp32_path": float(hrel.mean()),
                        "parity_gated": False}
            del hreplay, hout

    # The same forward with every conv on the fp32 MFMAs (KBN_NO_SPLIT=1: Winograd / 9-product up-convs / fused KB kernels,
    # round 2's v20 path): what the split-operand arithmetic buys, measured by the same driver run.
    fp32_only_fps = None
    if not args.no_fp32_mfma and not args.eager:
        with knob_env("KBN_NO_SPLIT"):
            freplay = model.capture(*frames, branches=args.branches or None)
            fp32_only_fps, _ = replay_rate(freplay, freplay.static_in, 10, 3, per, world, dev)
            del freplay

    # Side figure, NOT the headline: the reference's graph computes one conv whose result nothing reads -- conv_image of the
    # last KB level (src/networks.py:475-523: conv5_image takes conv4_fused) -- and so does the timed forward above.
    # KBNetEncoder.skip_unused_image leaves that launch out; the depth maps are the same bits.
    dead_conv_fps = None
    if not args.no_fp32_mfma and not args.eager and hasattr(model.encoder, "skip_unused_image"):
        model.encoder.skip_unused_image = True
        try:
            dreplay = model.capture(*frames, branches=args.branches or None)
            dout = dreplay(*dreplay.static_in)
            same_bits = bool(torch.equal(dout, forward(*frames)))
            drate, _ = replay_rate(dreplay, dreplay.static_in, 10, 3, per, world, dev)
            dead_conv_fps = {"frames_per_s": round(drate, 1),
                             "same_bits_as_timed_forward": same_bits,
                             "what": "conv_image of KB level 3 not launched (its output feeds nothing: reference src/networks.py:475-523); "
                                     "KBNetEncoder.skip_unused_image, off by default and in `value`"}
            del dreplay, dout
        finally:
            model.encoder.skip_unused_image = False

    # BASELINE configs[4] in miniature (8 frames of each of three shapes per rank and step, two weight sets, per-frame intrinsics)
    mixed_fps = None
    if not args.no_mixed and not args.eager:
        mixed_fps = mixed_stream_rate(dev, rank, world)

    ms_per_step = 1e3 * elapsed / args.steps
    fps = per * world * args.steps / elapsed
    value_source = f"{args.steps} timed steps"
    if sustained is not None:
        sustained["ratio_to_timed_steps"] = round(sustained["frames_per_s"] / fps, 4)
        if sustained["frames_per_s"] < 0.97 * fps:   # the short region flattered the part: the sustained figure is the headline
            value_source = f"sustained run of {sustained['steps']} steps (more than 3 % below the {args.steps} timed steps: {fps:.1f} frames/s)"
            fps = sustained["frames_per_s"]
            ms_per_step = 1e3 * per * world / fps
    gflop_frame = conv_gflop_per_frame(cfg, HEIGHT, WIDTH)
    multi_gpu = None
    if not args.eager:
        multi_gpu = multi_gpu_report(forward, step_inputs, out, per, world, rank, dev, fps, local_seconds, args.steps)

    # ---- roofline of the dominant kernel (this rank's launches; rank 0 prints) ----
    groups = summarise_profile(prof, args.steps)
    breakdown = {k: {"launches": g["launches"], "ms_total": round(g["seconds"] * 1e3, 3),
                     "avg_us": round(g["seconds"] / g["launches"] * 1e6, 2)} for k, g in groups.items()}
    conv_groups = {k: g for k, g in groups.items() if k.startswith("conv_")}
    dom = max(conv_groups, key=lambda k: conv_groups[k]["seconds"])
    d = conv_groups[dom]
    dwork, dtime, dlaunch, dexec = d["work"], d["seconds"], d["launches"], d["executed"]
    # SURVEY 8(d) / the bench contract: `achieved` = ALGORITHMIC FLOPs of the dominant kernel's launches (the reference's
    # direct-conv formulation: 2 * N * Hout * Wout * Cin * 9 * Cout per launch) / their time, `peak` = the dense peak of the pipe its
    # MFMAs run on, `frac` = achieved / peak.  A split-operand kernel spends `products_per_fp32_product` = 3 fp16 MFMA products on
    # every fp32 product, so its `frac` cannot exceed `ceiling_frac` = 1/3.  What the kernel really ISSUES (3 x the products plus the
    # tile padding it does not skip; = SQ_INSTS_MFMA x 32768 of the PMC collection) is reported beside it under `issued`.
    peak = kb.ops.PIPE_PEAK_TFLOPS[d["pipe"]]
    products = kb.ops.PIPE_PRODUCTS[d["pipe"]]
    achieved = dwork / dtime / 1e12
    issued = dexec / dtime / 1e12
    mfma_groups = [g for g in groups.values() if g["pipe"] is not None and g["has_executed"] and g["executed"] > 0]
    psec = pipe_seconds(groups)
    alg_sec = sum(kb.ops.PIPE_PRODUCTS[g["pipe"]] * g["work"] / (kb.ops.PIPE_PEAK_TFLOPS[g["pipe"]] * 1e12) for g in mfma_groups)
    roofline = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 3), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "algorithmic_frac": round(achieved / peak, 4),
                "products_per_fp32_product": products, "ceiling_frac": round(1.0 / products, 4),
                "frac_of_ceiling": round(achieved / peak * products, 4),
                "traffic": None,
                "launches": dlaunch, "avg_launch_us": round(dtime / dlaunch * 1e6, 2),
                "flop_per_launch": dwork / dlaunch,
                "issued": {"tflops": round(issued, 3), "frac": round(issued / peak, 4), "flop_per_launch": dexec / dlaunch,
                           "over_algorithmic": round(dexec / dwork, 4),
                           "what": "FLOPs of the MFMA instructions the kernel issues (products x algorithmic + tile padding)"},
                "algorithmic_multiple_of_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                # whole forward: time the ALGORITHMIC products of every MFMA launch of a step need at the dense peak of THEIR pipe
                # (x 3 on the fp16 matrix core for the split-operand kernels, x 1 on the fp32 MFMAs) / the timed step; and the same
                # for the FLOPs the launches issue.  Fractions, <= 1
                "whole_forward_frac": round(alg_sec / args.steps / (ms_per_step * 1e-3), 4),
                "whole_forward_issued_frac": round(psec / args.steps / (ms_per_step * 1e-3), 4),
                "whole_forward_executed_gflop_per_frame": round(sum(g["executed"] for g in mfma_groups) / args.steps / per / 1e9, 3),
                "whole_forward_algorithmic_multiple_of_peak":
                    round(fps / world * gflop_frame / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4),
                "mfma_kernels_eager_frac": round(psec / sum(g["seconds"] for g in mfma_groups), 4),
                "per_kernel": per_kernel_table(groups, args.steps)}
    if d["pipe"] == "fp16":
        roofline["pipe"] = ("fp16 MFMA (matrix core): every fp32 product is taken as three fp16 products over two-term "
                            "splits of both operands, fp32 accumulation (csrc/conv_split.hip); `achieved` / `frac` price the "
                            "reference's fp32 multiply-adds against this pipe's dense peak (ceiling 1/3), `issued` counts the fp16 "
                            "MFMA FLOPs that run.  A whole-chip stream of nothing but its instruction, "
                            "v_mfma_f32_32x32x16_f16, sustains 1.22 PFLOP/s on random operands from registers and 1.69 "
                            "interleaved with the LDS reads that feed it (profiles/r02/mfma_power_probe.txt)")
    if dom == "conv_wino":
        roofline["algorithm"] = "Winograd F(2x2,3x3) in fp32: 4/9 of the algorithmic multiply-adds reach the MFMAs"
    roofline["measured_on"] = ("eager whole-batch launches, one kernel at a time, HIP events on the launch stream "
                               "(the timed graph may run the batch as concurrent sub-batch branches; "
                               "`rocprofv3 --stats -- bench.py --branches 1` shows the same launches)")
    roofline["traffic"], roofline["pmc"] = lookup_traffic(dom, dlaunch, per)
    # the two north-star kernels that are bounded by bytes: S2D and the KB layer's fused front (HBM GB/s against 8 TB/s)
    for key, names in (("s2d_hbm", ("s2d",)), ("kb_front_hbm", ("kb1_front", "kb1_depth_front", "s2d_depth_front"))):
        hit = [groups[nm] for nm in names if nm in groups and groups[nm]["has_bytes"]]
        if hit:
            nb, sec, ln = sum(g["nbytes"] for g in hit), sum(g["seconds"] for g in hit), sum(g["launches"] for g in hit)
            roofline[key] = {"achieved": round(nb / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(nb / sec / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_us": round(sec / ln * 1e6, 2),
                             "kernels": [nm for nm in names if nm in groups]}

    result = base_result(fps, world, args.steps, args.warmup, ms_per_step)
    result.update({
        "config": {"workload": f"KITTI 352x1216, batch {per}/GPU (BASELINE configs[2] fp32 leg = configs[3] per-GPU share), "
                               "fp32 tensors and accumulation, full KBNet forward in HIP (S2D + KB layers + MFMA convs + head; "
                               "the wide 3x3 convs take their fp32 products as three fp16 MFMAs over split operands), "
                               "random xavier weights",
                   # what "f32" means on this path: tensors, accumulators and results are fp32; the wide 3x3 convs form every
                   # fp32 product from two-term fp16 splits of both operands (three exact fp16 x fp16 MFMA products, 22+ bits of
                   # each operand), measured error vs fp64 no larger than an fp32 MFMA chain's; same 1e-4 parity gate as before
                   "arithmetic": "fp32 in / fp32 accumulate / fp32 out; 3x3 convs with >= 48 filters: fp32 products as 3 fp16 MFMAs "
                                 "over split operands (csrc/conv_split.hip), everything else fp32 MFMA / VALU",
                   "frames_per_gpu": per, "global_batch": per * world, "height": HEIGHT, "width": WIDTH,
                   "gflop_per_frame": round(gflop_frame, 3),
                   "parallelism": f"frames sharded over {world} rank(s), RCCL all-gather of outputs",
                   "n_ranks_seen": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                   "collective_backend": backend + (" (TEST MODE: all ranks share cuda:0, gloo collectives -- not a scaling point)" if backend == "gloo-cuda" else ""),
                   "gathered_frames": int(out.shape[0]), "gather_matches_local_forward_rank0": gather_ok,
                   "rccl_version": rccl_version(),
                   # per-rank rates, the collective alone, rank 0 alone and value / (N x rank 0 alone): an N-rank line that explains itself
                   "multi_gpu": multi_gpu,
                   "launch": "eager" if args.eager else
                             f"HIP graph replay, {getattr(forward, 'branches', 1)} concurrent sub-batch branch(es)"
                             + (", one graph per branch on its own stream" if getattr(forward, "split_graphs", False) else ""),
                   "value_source": value_source,
                   # the same runner / graph for >= 300 more steps and >= 3 s (power-capped part: steady state, not a burst)
                   "sustained": sustained,
                   # graph replay of ONE frame, beside the reference's README latency
                   "batch1": batch1,
                   "eager_ms_per_step_with_event_timing": round(eager_ms, 4),
                   "reference_style_region_ms_per_step": round(refstyle_ms, 4),
                   # side measurement: VOID preset, 480x640, same batch per GPU, forward only (no all-gather)
                   "void_480x640_frames_per_s": None if void_fps is None else round(void_fps, 1),
                   # side measurement: the reference's other architecture switches at the same batch and size, forward only
                   "reference_options_frames_per_s": options_fps,
                   # side measurement: BASELINE configs[1] (batch 8 per GPU), forward only
                   "batch8_frames_per_s": None if side_fps is None else round(side_fps, 1),
                   # ... and in the latency form sized for four-frame launches (split-K where four frames cannot fill the chip; another summation order, opt-in)
                   "batch8_latency_mode_frames_per_s": None if side_latency_fps is None else round(side_latency_fps, 1),
                   # side measurement: the forward without the one conv of the reference's graph whose result nothing reads
                   "unused_image_conv_skipped": dead_conv_fps,
                   # side measurement: the same batch with every conv on the fp32 MFMAs (KBN_NO_SPLIT=1), graph replay
                   "fp32_mfma_only_frames_per_s": None if fp32_only_fps is None else round(fp32_only_fps, 1),
                   # side measurement: the tuned split kernels in one-term mode (fp16 h1 w1 alone) -- throughput only
                   "fp16_one_term_leg": fp16_leg,
                   # side measurement: BASELINE configs[4] in miniature -- VOID 480x640 / NYUv2 416x576 / KITTI 352x1216 round robin, 8 frames of
                   # each per rank and step, per-frame intrinsics, two weight sets resident, one graph per shape, one gather per shape bucket
                   "mixed_shape_stream_frames_per_s": None if mixed_fps is None else round(mixed_fps, 1)},
        "roofline": roofline, "kernels": breakdown,
    })

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, ref = cpu_baseline(cfg, sds, [f[0:1].cpu() for f in frames], [f[:SIDE_BATCH].cpu() for f in frames])
        result["cpu_baseline"] = base
        got = out[0:1].cpu()  # frame 0 of rank 0
        result["parity"] = {"max_rel_err_vs_oracle": float(((got - ref).abs() / ref.abs()).max()),
                            "mae_vs_oracle_m": float((got - ref).abs().mean()), "tolerance": 1e-4}
    else:
        result["cpu_baseline"] = None   # rank 0 at N = 1 only (the contract): the host cores are shared by all ranks
    emit(result, rank)
    if world > 1:
        kb.dist.barrier()   # ranks leave together (rank 0 printed its line; nobody tears the group down under a peer's collective)
        torch.distributed.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
