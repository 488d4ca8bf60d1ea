#!/usr/bin/env python3
"""Per-layer micro-benchmark of kbn_conv2d_forward over forced tile choices (GPU box).
usage: conv_bench.py [layer-name-substring ...]   (env KBN_FORCE_MW / KBN_FORCE_TWB are set per run)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name: (cin list (sources), cout, k, stride, H_in, W_in, resize_from or None) at batch 8, KITTI
LAYERS = {
    "conv0_image": ([3], 48, 3, 1, 352, 1216, None),
    "conv0_depth": ([8], 16, 3, 1, 352, 1216, None),
    "kb1_image": ([48], 48, 3, 2, 352, 1216, None),
    "kb1_depth": ([16, 3], 16, 3, 2, 352, 1216, None),
    "kb1_fused": ([48, 3], 48, 1, 2, 352, 1216, None),
    "kb2_image": ([48], 96, 3, 2, 176, 608, None),
    "kb2_fused": ([48, 3, 48], 96, 1, 2, 176, 608, None),
    "kb3_image": ([96], 192, 3, 2, 88, 304, None),
    "kb4_image": ([192], 384, 3, 2, 44, 152, None),
    "conv5_image": ([384], 384, 3, 2, 22, 76, None),
    "deconv4_up": ([512], 256, 3, 1, 22, 76, (11, 38)),
    "deconv4_conv": ([256, 512], 256, 3, 1, 22, 76, None),
    "deconv3_up": ([256], 128, 3, 1, 44, 152, (22, 76)),
    "deconv3_conv": ([128, 256], 128, 3, 1, 44, 152, None),
    "deconv2_conv": ([128, 128], 128, 3, 1, 88, 304, None),
    "deconv1_up": ([128], 64, 3, 1, 176, 608, (88, 304)),
    "deconv1_conv": ([64, 64], 64, 3, 1, 176, 608, None),
    "deconv0_up": ([64], 12, 3, 1, 352, 1216, (176, 608)),
    "deconv0_conv": ([12], 12, 3, 1, 352, 1216, None),
}


def run_one(name, batch=int(os.environ.get("KBN_BATCH", "8")), iters=8):
    import torch
    import kbnet_amd as kb
    cins, cout, k, stride, h, w, rs = LAYERS[name]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    srcs, tens = [], []
    for c in cins:
        if c == 3 and len(cins) > 1:  # stand-in for coords/xyz: plain tensor of 3 channels
            t = torch.randn(batch, 3, h, w, generator=g).to(dev)
        else:
            hh, ww = rs if rs else (h, w)
            t = torch.randn(batch, c, hh, ww, generator=g).to(dev)
        tens.append(t)
        srcs.append(kb.ops.tensor_src(t))
    cin = sum(cins)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
    pw = kb.ops.pack_conv_weight(wt, stride)
    oh, ow = -(-h // stride), -(-w // stride)
    out = torch.empty(batch, cout, oh, ow, device=dev)
    f = lambda: kb.ops.conv2d(srcs, pw, batch, cout, k, stride, h, w, out, resize=rs is not None, negative_slope=0.2)
    if rs is not None and (h, w) == (2 * rs[0], 2 * rs[1]) and not os.environ.get("KBN_NO_UP2X"):
        pw2 = kb.ops.pack_upconv2x_weight(wt)
        f = lambda: kb.ops.upconv2x(tens[0], pw2, cout, out, 0.2)
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    samples = []
    for _ in range(5):   # median of 5 timing blocks (clock ramp / co-tenant noise)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            f()
        e.record()
        torch.cuda.synchronize()
        samples.append(s.elapsed_time(e) * 1e3 / iters)
    us = sorted(samples)[2]
    flops = 2.0 * batch * oh * ow * cin * k * k * cout
    pl = kb.ops.conv_plan(batch, cout, cin, k, stride, h, w, rs is not None)
    return {"layer": name, "us": round(us, 1), "tflops": round(flops / us / 1e6, 1), **pl}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        print(json.dumps(run_one(sys.argv[2])))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--ablate":
        for n in sys.argv[2:]:
            for dbg, tag in ((0, "full"), (1, "no-A-staging"), (2, "no-B-staging"), (3, "no-staging"), (4, "no-MFMA"), (7, "barriers-only"), (8, "A-from-cache"), (12, "A-from-cache-noMFMA")):
                env = dict(os.environ, KBN_DEBUG=str(dbg))
                r = subprocess.run([sys.executable, __file__, "--one", n], env=env, capture_output=True, text=True)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                d = json.loads(line[-1])
                print(f"{n:14s} {tag:14s} {d['us']:8.1f} us {d['tflops']:6.1f} TF-equivalent", flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--dbg":  # --dbg 0,1,2,3 layer...: KBN_DEBUG values, raw table
        for n in sys.argv[3:]:
            for dbg in sys.argv[2].split(","):
                env = dict(os.environ, KBN_DEBUG=dbg)
                r = subprocess.run([sys.executable, __file__, "--one", n], env=env, capture_output=True, text=True, timeout=120)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                d = json.loads(line[-1])
                print(f"{n:14s} dbg={dbg:3s} {d['us']:8.1f} us {d['tflops']:6.1f} TF-equivalent  {d['kernel']} wgs={d['workgroups']} MW={d['MW']} TWB={d['TWB']}", flush=True)
        sys.exit(0)
    pats = sys.argv[1:]
    names = [n for n in LAYERS if not pats or any(p in n for p in pats)]
    for n in names:
        best = None
        for mw in (0, 1, 2, 4, 8):
            for twb in ((0,) if mw == 0 else (1, 2, 4)):
                env = dict(os.environ, KBN_FORCE_MW=str(mw), KBN_FORCE_TWB=str(twb))
                r = subprocess.run([sys.executable, __file__, "--one", n], env=env, capture_output=True, text=True)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if not line:
                    continue
                d = json.loads(line[-1])
                if mw and (d["MW"] != mw or d["TWB"] != twb):
                    continue
                tag = "auto" if mw == 0 else f"MW{mw} TWB{twb}"
                print(f"{n:14s} {tag:10s} {d['us']:8.1f} us {d['tflops']:6.1f} TF  MW={d['MW']} TWB={d['TWB']} TH={d['TH']} wgs={d['workgroups']} pipe={d['pipelined']}", flush=True)
