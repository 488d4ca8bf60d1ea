#!/usr/bin/env python3
"""Throughput of the input pipeline (row f4) on the GPU box: KITTI-shaped PNG files (352 x 3648 RGB image
triplets, 352 x 1216 16-bit sparse depth) -> InferenceFrameLoader -> preprocess -> graph-replayed forward.
Prints decode-only, loader-only (decode + H2D + unpack) and end-to-end frames/s, next to the PIL loop the
reference runs (one worker, one sample at a time).   usage: loader_bench.py [n_files] [workers] [prefetch]"""
import os, sys, time, tempfile
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kbnet_amd as kb
from PIL import Image

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 64
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
prefetch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
H, W = 352, 1216
d = tempfile.mkdtemp(prefix="kbn_loader_")
g = np.random.Generator(np.random.Philox(5))
yy, xx = np.mgrid[0:H, 0:3 * W]
imgs, deps, ks = [], [], []
for i in range(n_files):   # smooth scene + sensor noise: compresses like a photograph (~2x), unlike white noise
    base = 128 + 60 * np.sin(xx / 90.0 + i) * np.cos(yy / 40.0) + 30 * np.sin((xx + yy) / 17.0)
    rgb = np.stack([base + g.normal(0, 6, base.shape) + 20 * c for c in range(3)], axis=-1).clip(0, 255).astype(np.uint8)
    p = os.path.join(d, f"im{i}.png"); Image.fromarray(rgb).save(p, compress_level=6); imgs.append(p)
    z = (g.random((H, W)) < 0.05) * (g.random((H, W)) * 79 + 1) * 256
    p = os.path.join(d, f"sd{i}.png"); Image.fromarray(z.astype(np.uint16)).save(p); deps.append(p)
    p = os.path.join(d, f"k{i}.npy"); np.save(p, np.array([[721.5, 0, 609.6], [0, 721.5, 172.9], [0, 0, 1]])); ks.append(p)
mb = sum(os.path.getsize(p) for p in imgs + deps) / 1e6
print(f"{n_files} samples, {mb / n_files:.2f} MB of PNG per sample, {workers} loader threads, {os.cpu_count()} host cores")

dev = torch.device("cuda:0")
t0 = time.perf_counter()
for i in range(min(n_files, 16)):   # the reference's loop: PIL, one sample at a time (src/datasets.py:259-283)
    im = np.asarray(Image.open(imgs[i]).convert("RGB"), np.float32).transpose(2, 0, 1)
    _, im, _ = np.split(im, 3, axis=-1)
    z = np.array(Image.open(deps[i]), dtype=np.float32) / 256.0
    k = np.load(ks[i]).astype(np.float32)
    torch.from_numpy(np.ascontiguousarray(im)).to(dev); torch.from_numpy(z).to(dev); torch.from_numpy(k).to(dev)
torch.cuda.synchronize()
print(f"reference-style PIL loop, 1 worker : {min(n_files, 16) / (time.perf_counter() - t0):8.1f} frames/s")

def run_loader(consume):
    loader = kb.loader.InferenceFrameLoader(imgs, deps, ks, use_image_triplet=True, batch_size=8, device=dev, workers=workers, prefetch=prefetch)
    for batch in loader:   # warm (pinned buffers, page cache)
        consume(*batch)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        for batch in loader:
            consume(*batch)
    torch.cuda.synchronize()
    return 3 * n_files / (time.perf_counter() - t)

print(f"loader (decode + H2D + unpack)     : {run_loader(lambda *b: None):8.1f} frames/s")
cfg = kb.kitti_config()
model = kb.modules.KBNetModel.from_config(cfg, dev)
model.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN["kitti"]))
frames = [t.to(dev) for t in kb.synthetic.make_frames(8, H, W, "kitti", seed=1)]
replay = model.capture(*frames)
def step(image, sparse, k):
    img, valid, filt = kb.ops.preprocess(image, sparse)
    if image.shape[0] == 8:
        replay(img, filt, valid, k)
    else:
        model.forward(img, filt, valid, k)
print(f"files -> depth maps, end to end    : {run_loader(step):8.1f} frames/s")
