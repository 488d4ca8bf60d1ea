#!/usr/bin/env python3
"""Parity margin report (GPU box; lives under tests/ because it runs the oracle).

For every preset (KITTI 352x1216, VOID 480x640, NYUv2 416x576) and every weight / input seed, ONE line with the worst
element-wise relative error of the depth map

    shipped HIP path  vs the fp32 oracle | vs an fp64 evaluation of the network
    KBN_NO_SPLIT=1    vs the fp32 oracle | vs fp64          (every conv on the fp32 MFMAs: no split operands, no pair tensors)
    fp32 oracle       vs fp64

and the largest window slack of the forward's pair tensors (binades between the top of the fp16 window a producer chose
from its BOUND and the maximum it then measured).  Two fp32 evaluation orders of a 35-conv network cannot agree better
with each other than each agrees with the exact result: the fp64 columns are what discriminates.

usage: parity_margin.py [--seeds N] [--trained] [--out FILE] [--presets kitti,void,nyu_v2] [--deconv-type transpose] [--activation elu] [--latency-frames 1]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import kbnet_amd as kb
from oracle import kbnet_oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=32)
ap.add_argument("--trained", action="store_true", help="trained-like weight statistics (synthetic.make_state_dicts(trained_like=True))")
ap.add_argument("--out", default=None)
ap.add_argument("--presets", default="kitti,void,nyu_v2")
ap.add_argument("--deconv-type", default="up", help="run_kbnet.py --deconv_type (up | transpose)")
ap.add_argument("--activation", default="leaky_relu", help="run_kbnet.py --activation_func (leaky_relu | relu | elu | sigmoid | linear)")
ap.add_argument("--latency-frames", type=int, default=0, help="the shipped-path columns in the latency form: KBNetModel.set_latency_mode(True, frames=N)")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.set_num_threads(min(16, os.cpu_count() or 1))
SHAPES = {"kitti": (352, 1216), "void": (480, 640), "nyu_v2": (416, 576)}
out_f = open(args.out, "w") if args.out else None


def say(line):
    print(line, flush=True)
    if out_f:
        out_f.write(line + "\n")
        out_f.flush()


def worst(a, b):
    return float(((a.double() - b.double()).abs() / b.double().abs()).max())


def hip_forward(cfg, sds, frames, no_split):
    if no_split:
        os.environ["KBN_NO_SPLIT"] = "1"
    else:
        os.environ.pop("KBN_NO_SPLIT", None)
    kb.ops.reload_env()
    try:
        m = kb.modules.KBNetModel.from_config(cfg, dev)
        m.load_state_dicts(*sds)
        if args.latency_frames and not no_split:
            m.set_latency_mode(True, frames=args.latency_frames)
        kb.ops.PairTensor.LOG = log = []
        out = m.forward(*[f.to(dev) for f in frames]).cpu()
        slack = max((float(t.window_slack_log2().max()) for t in log), default=float("nan"))
        return out, slack, len(log)
    finally:
        kb.ops.PairTensor.LOG = None
        os.environ.pop("KBN_NO_SPLIT", None)
        kb.ops.reload_env()


say(f"# parity margin: {args.seeds} seeds x presets {args.presets}; {'LATENCY FORM (frames = %d); ' % args.latency_frames if args.latency_frames else ''}deconv_type {args.deconv_type}, activation {args.activation}; weights {'trained-like (t3 entries, 2^7 filter spread, 10 % dead)' if args.trained else 'xavier'}; "
    f"device {torch.cuda.get_device_name(0)}; columns: max element-wise relative error of the depth map")
say("# preset seed | hip_vs_oracle hip_vs_fp64 | nosplit_vs_oracle nosplit_vs_fp64 | oracle_vs_fp64 | max pair-window slack (binades), pair tensors")
t_all = time.time()
for preset in args.presets.split(","):
    shape = SHAPES[preset]
    import dataclasses
    cfg = dataclasses.replace(kb.PRESETS[preset](), deconv_type=args.deconv_type, activation_func=args.activation)
    slope = orc.activation_slope(args.activation)
    rows = []
    for seed in range(args.seeds):
        sds = kb.synthetic.make_state_dicts(cfg, seed=seed, gain=kb.synthetic.PARITY_GAIN[preset], trained_like=args.trained)
        frames = kb.synthetic.make_frames(1, *shape, preset, seed=1 + seed, jitter_intrinsics=0.1)
        a = (cfg.min_pools, cfg.max_pools, cfg.min_predict_depth, cfg.max_predict_depth)
        ref = orc.kbnet_forward(*frames, *sds, *a, slope=slope)
        torch.set_default_dtype(torch.float64)      # the oracle's pixel grid follows the default dtype (reference quirk Q8)
        try:
            ref64 = orc.kbnet_forward(*[f.double() for f in frames], *[{k: v.double() for k, v in sd.items()} for sd in sds], *a, slope=slope)
        finally:
            torch.set_default_dtype(torch.float32)
        out, slack, npair = hip_forward(cfg, sds, frames, no_split=False)
        out_ns, _, _ = hip_forward(cfg, sds, frames, no_split=True)
        row = (worst(out, ref), worst(out, ref64), worst(out_ns, ref), worst(out_ns, ref64), worst(ref, ref64), slack)
        rows.append(row)
        flag = "  <-- above the 1e-4 gate vs the oracle" if row[0] >= 1e-4 else ""
        say(f"{preset:7s} {seed:3d} | {row[0]:.3e} {row[1]:.3e} | {row[2]:.3e} {row[3]:.3e} | {row[4]:.3e} | {slack:5.2f} {npair}{flag}")
    col = lambda i: [r[i] for r in rows]
    say(f"{preset:7s} worst of {args.seeds} | {max(col(0)):.3e} {max(col(1)):.3e} | {max(col(2)):.3e} {max(col(3)):.3e} | {max(col(4)):.3e} | "
        f"{max(col(5)):5.2f}   (gate 1e-4; seeds with hip_vs_fp64 > 2 x oracle_vs_fp64: "
        f"{[s for s, r in enumerate(rows) if r[1] > 2 * r[4]]}; nosplit: {[s for s, r in enumerate(rows) if r[3] > 2 * r[4]]})")
say(f"# {time.time() - t_all:.0f} s")
