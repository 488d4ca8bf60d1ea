#!/usr/bin/env python3
"""CPU experiment for the round-6 price-out of a 1-D Winograd F(2,3) form of the decoder's concat convs ON SPLIT OPERANDS
(profiles/r06/wino1d_priceout.txt): what the transform costs in accuracy before anything is built.

Direct form (csrc/conv_split.hip):  y = sum_c sum_taps a w, every fp32 product taken as h1 w1 + 2^-11 (h1 w2' + h2 w1).
F(2,3) along the row:  two outputs of a row from four inputs d0..d3 with four products per (channel, filter, tap row):
    V = B^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3),   U = G g = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2),
    y0 = M0 + M1 + M2,  y1 = M1 - M2 - M3,  M = U * V summed over channels and the three tap rows.
V is formed in fp32 (one rounding per element), then split like an activation (window on max |V| <= 2 max |d|); U is formed in
fp64 at pack time, rounded to fp32 and split like a weight (exponent per filter).  Products are exact; accumulation is taken in
fp64 here, so the figures isolate the REPRESENTATION error of each form (the GPU's fp32 accumulation adds the same to both).
usage: winograd1d_split_error.py [trials]"""
import math, sys
import numpy as np

rng = np.random.default_rng(5)


def split_act(a, amax=None):
    amax = float(np.abs(a).max()) if amax is None else amax
    k = 14 - math.floor(math.log2(amax)) if amax > 0 else 0
    ap = (a.astype(np.float32) * np.float32(2.0 ** k)).astype(np.float32)
    h1 = ap.astype(np.float16)
    h1 = np.where(np.abs(h1.astype(np.float32)) < 2.0 ** -14, np.float16(0), h1)
    h2 = ((ap - h1.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return h1.astype(np.float64), h2.astype(np.float64), k


def split_w(w):   # w: [filters, ...]; exponent per filter
    flat = w.reshape(w.shape[0], -1)
    e = np.array([13 - math.frexp(float(np.abs(r).max()))[1] if np.abs(r).max() > 0 else 0 for r in flat])
    ws = (w.astype(np.float32) * (2.0 ** e).astype(np.float32).reshape((-1,) + (1,) * (w.ndim - 1))).astype(np.float32)
    w1 = ws.astype(np.float16)
    w2 = ((ws - w1.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    fl = lambda x: np.where(np.abs(x.astype(np.float32)) < 2.0 ** -14, np.float16(0), x)
    return w1.astype(np.float64), fl(w2).astype(np.float64), e


def one(cin, cout, width=64):
    # one output row segment: inputs d[c][3 rows][width + 2], weights g[o][c][3][3]; LeakyReLU'd gaussian activations
    d = rng.standard_normal((cin, 3, width + 2)).astype(np.float32)
    d = np.where(d < 0, 0.2 * d, d).astype(np.float32)
    g = (rng.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
    d64, g64 = d.astype(np.float64), g.astype(np.float64)
    truth = np.zeros((cout, width))
    for ky in range(3):
        for kx in range(3):
            truth += np.einsum("oc,cx->ox", g64[:, :, ky, kx], d64[:, ky, kx:kx + width])
    # fp32 operands, exact products (the floor both forms share): nothing to add, operands are fp32 already -> 0
    # direct split form
    h1, h2, k = split_act(d)
    w1, w2, e = split_w(g)
    acc = np.zeros((cout, width))
    for ky in range(3):
        for kx in range(3):
            a1, a2 = h1[:, ky, kx:kx + width], h2[:, ky, kx:kx + width]
            acc += np.einsum("oc,cx->ox", w1[:, :, ky, kx], a1) + 2.0 ** -11 * (np.einsum("oc,cx->ox", w2[:, :, ky, kx], a1) + np.einsum("oc,cx->ox", w1[:, :, ky, kx], a2))
    direct = acc * (2.0 ** -(e + k)).reshape(-1, 1)
    # F(2,3) along x on split operands
    t = width // 2
    dd = np.stack([d[:, :, 2 * i:2 * i + 4] for i in range(t)], axis=2)          # c, row, tile, 4   (fp32)
    V = np.stack([dd[..., 0] - dd[..., 2], dd[..., 1] + dd[..., 2], dd[..., 2] - dd[..., 1], dd[..., 1] - dd[..., 3]], axis=-1).astype(np.float32)
    U64 = np.stack([g64[..., 0], (g64[..., 0] + g64[..., 1] + g64[..., 2]) / 2, (g64[..., 0] - g64[..., 1] + g64[..., 2]) / 2, g64[..., 2]], axis=-1)
    U = U64.astype(np.float32)                                                     # o, c, ky, 4
    v1, v2, kv = split_act(V)
    u1, u2, eu = split_w(U)
    M = np.zeros((cout, t, 4))
    for ky in range(3):
        M += np.einsum("ocf,ctf->otf", u1[:, :, ky], v1[:, ky]) + 2.0 ** -11 * (np.einsum("ocf,ctf->otf", u2[:, :, ky], v1[:, ky]) + np.einsum("ocf,ctf->otf", u1[:, :, ky], v2[:, ky]))
    M *= (2.0 ** -(eu + kv)).reshape(-1, 1, 1)
    y = np.empty((cout, width))
    y[:, 0::2] = M[..., 0] + M[..., 1] + M[..., 2]
    y[:, 1::2] = M[..., 1] - M[..., 2] - M[..., 3]
    # the same transform with UNSPLIT fp32 operands (V rounded to fp32, U rounded to fp32): the part of the error that is Winograd's own
    M32 = np.zeros((cout, t, 4))
    for ky in range(3):
        M32 += np.einsum("ocf,ctf->otf", U[:, :, ky].astype(np.float64), V[:, ky].astype(np.float64))
    y32 = np.empty((cout, width))
    y32[:, 0::2] = M32[..., 0] + M32[..., 1] + M32[..., 2]
    y32[:, 1::2] = M32[..., 1] - M32[..., 2] - M32[..., 3]
    rms = math.sqrt((truth ** 2).mean())
    r = lambda z: (math.sqrt(((z - truth) ** 2).mean()) / rms, float(np.abs(z - truth).max()) / rms)
    return r(direct), r(y), r(y32)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    print("representation error against fp64 (rms / max, in units of the output's rms), exact products, fp64 accumulation")
    for name, cin, cout in (("deconv1 conv 128 -> 64", 128, 64), ("deconv2 conv 256 -> 128", 256, 128), ("deconv4 conv 768 -> 256", 768, 256)):
        rows = np.array([one(cin, cout) for _ in range(trials)])      # trials, 3 forms, 2
        m = rows.mean(axis=0)
        print(f"{name:26s} direct split {m[0][0]:.2e} / {m[0][1]:.2e}   F(2,3) split {m[1][0]:.2e} / {m[1][1]:.2e}   "
              f"F(2,3) fp32 operands {m[2][0]:.2e} / {m[2][1]:.2e}   ratio split forms {m[1][0] / m[0][0]:.2f}")
    print("for scale: an fp32 fmaf chain over K = 1152 .. 6912 products leaves 4.4e-7 .. 1.7e-6 rms (profiles/r02/bf16x_probe.txt)")


if __name__ == "__main__":
    main()
