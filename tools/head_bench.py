#!/usr/bin/env python3
"""Times the fused decoder tail (kbn_conv_head_forward: deconv0's conv + output0 + depth mapping) against the
two-launch path, KITTI 352x1216, 12 channels, under the kernel's KBN_DEBUG phase ablations (GPU box)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch, kbnet_amd as kb
    n = int(os.environ.get("KBN_BATCH", "16"))
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, 12, 352, 1216, generator=g).to(dev)
    wc = (torch.randn(12, 12, 3, 3, generator=g) * 0.1).to(dev)
    wo = (torch.randn(1, 12, 3, 3, generator=g) * 0.3).to(dev)
    out = torch.empty(n, 1, 352, 1216, device=dev)
    if sys.argv[2] == "fused":
        f = lambda: kb.ops.conv_head(x, wc, wo, 1.5, 100.0, 0.2, out=out)
    else:
        pw = kb.ops.pack_conv_weight(wc, 1)
        mid = torch.empty_like(x)
        def f():
            kb.ops.conv2d([kb.ops.tensor_src(x)], pw, n, 12, 3, 1, 352, 1216, mid, negative_slope=0.2)
            kb.ops.depth_head(mid, wo, 1.5, 100.0, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    print(json.dumps({"us": round(s.elapsed_time(e) * 100, 1)}))
    sys.exit(0)
for mode, dbg, tag in (("split", 0, "two launches"), ("fused", 0, "fused"), ("fused", 1, "fused, no staging"), ("fused", 2, "fused, no MFMA"),
                       ("fused", 4, "fused, no exchange/head"), ("fused", 6, "fused, staging + stores only"), ("fused", 7, "fused, skeleton")):
    r = subprocess.run([sys.executable, __file__, "--one", mode], env=dict(os.environ, KBN_DEBUG=str(dbg)), capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(f"{tag:30s}", line[-1] if line else r.stderr[-300:], flush=True)
