#!/usr/bin/env python3
"""Debug aid for kbn_s2d_depth_front_forward (lives under tests/ because it runs the oracle): with conv0_depth / conv_depth set to centre-tap identities the launch returns the
on-chip S2D tensor at the even pixels (through two LeakyReLUs, undone here); compared channel by channel with the oracle's S2D."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kbnet_amd as kb
from oracle import kbnet_oracle as orc
dev = torch.device("cuda:0")
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 96)
mode = sys.argv[3] if len(sys.argv) > 3 else "full"
cfg = kb.kitti_config()
mins, maxs = list(cfg.min_pools), list(cfg.max_pools)
g = torch.Generator().manual_seed(3)
n, nf = 1, 8
npool = len(mins) + len(maxs)
mask = (torch.rand(n, 1, h, w, generator=g) < 0.3).float()
z = torch.round((1.0 + 79.0 * torch.rand(n, 1, h, w, generator=g)) * 256.0) / 256.0 * mask
x = torch.cat([z, (z > 0).float()], 1)
sd = {"pool_convs.0.conv.weight": torch.randn(nf, npool, 1, 1, generator=g) / npool ** 0.5,
      "pool_convs.1.conv.weight": torch.randn(nf, nf, 1, 1, generator=g) / nf ** 0.5,
      "pool_convs.2.conv.weight": torch.randn(nf, nf, 1, 1, generator=g) / nf ** 0.5,
      "conv.conv.weight": torch.randn(nf, nf + 2, 3, 3, generator=g) / ((nf + 2) * 9) ** 0.5}
if mode == "chain":      # the 3x3 conv passes feature channel f through its centre tap: exposes the 1x1 chain
    sd["conv.conv.weight"].zero_()
    for f in range(8):
        sd["conv.conv.weight"][f, f, 1, 1] = 1.0
if mode in ("pool", "l0", "l1", "l2"):   # identity layers except one: exposes the pooled values / a single chain layer
    sd["conv.conv.weight"].zero_()
    for f in range(8):
        sd["conv.conv.weight"][f, f, 1, 1] = 1.0
    for i in range(3):
        if mode != f"l{i}":
            wt = sd[f"pool_convs.{i}.conv.weight"]
            wt.zero_()
            for f in range(min(wt.shape[0], wt.shape[1])):
                wt[f, f, 0, 0] = 1.0
if mode == "raw":        # only the raw channels
    sd["conv.conv.weight"][:, :8] = 0
if mode == "feat":
    sd["conv.conv.weight"][:, 8:] = 0
w0 = torch.zeros(16, 8, 3, 3)
for f in range(16):
    w0[f, f % 8, 1, 1] = 1.0
wc = torch.zeros(16, 19, 3, 3)
for f in range(16):
    wc[f, f, 1, 1] = 1.0
proj = torch.zeros(1, 16, 1, 1); proj[0, 0] = 1.0
kmat = torch.tensor([[[60.0, 0.0, w / 2.0], [0.0, 58.0, h / 2.0], [0.0, 0.0, 1.0]]]).repeat(n, 1, 1)
ref = orc.sparse_to_dense_pool(x, sd, mins, maxs)
packed_s = kb.ops.pack_s2d_depth_front_weight([sd[f"pool_convs.{i}.conv.weight"].to(dev) for i in range(3)], sd["conv.conv.weight"].to(dev))
packed_d = kb.ops.pack_kb1_depth_front_weight(w0.to(dev), wc.to(dev), proj.to(dev))
kinv = kb.ops.intrinsics_inverse(kmat.to(dev))
oh, ow = (h + 1) // 2, (w + 1) // 2
out_d = torch.zeros((n, 16, oh, ow), device=dev)
res = kb.ops.s2d_depth_front(x.to(dev), kinv, packed_s, packed_d, mins, maxs, 16, 16, out_d, 0.2, 0.2, 0.2, 0.2)
got = out_d.cpu()
inv = lambda t: torch.where(t > 0, t, t / 0.2)
got = inv(inv(got))[:, :8]
want = ref[:, :, ::2, ::2]
err = (got - want).abs()
scale = want.abs().amax(dim=(2, 3), keepdim=True).clamp_min(1e-30)
print("per-channel max rel err:", [f"{float(e):.1e}" for e in (err / scale).amax(dim=(0, 2, 3))])
bad = (err / scale) > 1e-3
print("bad pixels per channel:", bad.sum(dim=(0, 2, 3)).tolist(), "of", oh * ow)
if bad.any():
    idx = bad.nonzero()
    print("first bad (n, c, y, x):", idx[:12].tolist())
    ys, xs = idx[:, 2], idx[:, 3]
    print("bad y range", int(ys.min()), int(ys.max()), "x range", int(xs.min()), int(xs.max()))
    print("bad by x % 16:", torch.bincount(xs % 16, minlength=16).tolist())
    print("bad by y % 8:", torch.bincount(ys % 8, minlength=8).tolist())
    c, y, xx = idx[0, 1], idx[0, 2], idx[0, 3]
    print("got", float(got[0, c, y, xx]), "want", float(want[0, c, y, xx]))
    cb = int((err / scale).amax(dim=(0, 2, 3)).argmax())
    print("worst channel", cb, "row 5 got ", [round(float(v), 4) for v in got[0, cb, 5, :12]])
    print("worst channel", cb, "row 5 want", [round(float(v), 4) for v in want[0, cb, 5, :12]])
