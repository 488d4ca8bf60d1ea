#!/usr/bin/env python3
"""Raw throughput of kbn_png_decode_batch on KITTI-shaped image triplets (host only).
usage: decode_bench.py [threads ...]"""
import io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kbnet_amd as kb
from PIL import Image
H, W = 352, 3648
g = np.random.Generator(np.random.Philox(5))
yy, xx = np.mgrid[0:H, 0:W]
base = 128 + 60 * np.sin(xx / 90.0) * np.cos(yy / 40.0) + 30 * np.sin((xx + yy) / 17.0)
rgb = np.stack([base + g.normal(0, 6, base.shape) + 20 * c for c in range(3)], axis=-1).clip(0, 255).astype(np.uint8)
b = io.BytesIO(); Image.fromarray(rgb).save(b, "PNG", compress_level=6)
data = b.getvalue()
print(f"{len(data) / 1e6:.2f} MB PNG, {rgb.nbytes / 1e6:.2f} MB raw, {os.cpu_count()} cores")
n = 128
files = [data] * n
outs = [np.empty_like(rgb) for _ in range(n)]
for t in [int(a) for a in sys.argv[1:]] or [1, 8, 16, 32, 64, 128]:
    kb.loader.decode_png_batch(files[:t], outs[:t], threads=t)
    t0 = time.perf_counter()
    kb.loader.decode_png_batch(files, outs, threads=t)
    dt = time.perf_counter() - t0
    print(f"{t:4d} threads: {n / dt:8.1f} images/s  ({dt / n * t * 1e3:.1f} ms per image per thread)")
