#!/usr/bin/env python3
"""Checks candidate tile cost models (choose_tile in csrc/conv_igemm.hip) against a measured sweep
written by tools/conv_bench.py: prints, per layer, the tile each model picks and its regret
(time of the pick / best time in the sweep - 1).
usage: tile_model.py gpurun_out/conv_bench2.txt"""
import re, sys, math
from conv_bench import LAYERS


def plan(oc, cin, ks, stride):
    ck = 16 if ks == 1 else (4 if (cin <= 4 or stride == 2) else 8)
    nblk = -(-oc // 16)
    best, bestpad = 1, 1 << 30
    for nb in range(1, 5):
        pad = -(-nblk // nb) * nb
        if pad < bestpad or (pad == bestpad and nb > best):
            best, bestpad = nb, pad
    return ck, best, -(-nblk // best)


def model_old(mw, twb, L, n=8):
    cins, oc, ks, stride, h, w, rs = L
    ck, nb, ntn = plan(oc, sum(cins), ks, stride)
    oh, ow = -(-h // stride), -(-w // stride)
    s2 = ks == 3 and stride == 2
    th, tw = 4 * mw // twb, twb * 16
    tiles = -(-ow // tw) * -(-oh // th) * n * ntn
    rows = th if ks == 1 else (2 * th + 1 if s2 else th + 2)
    cols = stride * tw if ks == 1 else (2 * tw + 4 if s2 else tw + 8)
    lds = 2.0 * 4.0 * (ck * (rows * cols + 32.0) + ck * ks * ks * nb * 16)
    resident = int(160 * 1024 / lds)
    pen = 1.0 if resident >= 2 else 1.18
    return -(-tiles // 256) * (mw * 64.0 + 12.0 + 0.02 * rows * cols) * pen


def model_new(mw, twb, L, n=8):
    cins, oc, ks, stride, h, w, rs = L
    ck, nb, ntn = plan(oc, sum(cins), ks, stride)
    oh, ow = -(-h // stride), -(-w // stride)
    s2 = ks == 3 and stride == 2
    th, tw = 4 * mw // twb, twb * 16
    tiles = -(-ow // tw) * -(-oh // th) * n * ntn
    rows = th if ks == 1 else (2 * th + 1 if s2 else th + 2)
    cols = stride * tw if ks == 1 else (2 * tw + 4 if s2 else tw + 8)
    lds = 2.0 * 4.0 * (ck * (rows * cols + 32.0) + ck * ks * ks * nb * 16)
    resident = min(2, int(160 * 1024 / lds))
    if resident < 1:
        return 1e30
    slots = 256 * resident
    # per-workgroup time when `resident` workgroups share a CU's MFMA pipes
    mfma = mw * nb * 16.0                       # MFMA issue per chunk-tap group, arbitrary unit
    stage = 0.012 * rows * cols * (4.0 / nb)    # staging bytes per MFMA unit grow as NB shrinks
    wg = resident * (mfma + 6.0) + stage
    if resident == 1:
        wg *= 1.18
    full, tail = divmod(tiles, slots)
    t = full * wg
    if tail:
        t += wg * (0.6 + 0.4 * tail / slots)    # a thin last wave runs faster than a full one
    return t


def main(path):
    meas = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+(auto|MW(\d) TWB(\d))\s+([\d.]+) us", line)
        if not m:
            continue
        name = m.group(1)
        if m.group(2) == "auto":
            continue
        meas.setdefault(name, {})[(int(m.group(3)), int(m.group(4)))] = float(m.group(5))
    for name, tab in meas.items():
        L = LAYERS[name]
        if L[6] is not None:
            continue
        best = min(tab.values())
        out = [f"{name:14s} best {min(tab, key=tab.get)} {best:7.1f}"]
        for mdl in (model_old, model_new):
            pick = min(tab, key=lambda k: mdl(k[0], k[1], L))
            out.append(f"{mdl.__name__} {pick} +{100 * (tab[pick] / best - 1):.1f}%")
        print("  ".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
