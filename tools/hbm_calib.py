#!/usr/bin/env python3
"""HBM calibration on this box: torch fill / copy / read-reduce rates at the sizes the full-resolution layers move
(conv0 writes 876 MB per 8 KITTI frames)."""
import torch, time
dev = torch.device("cuda:0")
n = 876 * 1024 * 1024 // 4
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - s) / reps
dt = t(lambda: a.fill_(1.0)); print("fill  (write only): %.0f us  %.2f TB/s" % (dt * 1e6, n * 4 / dt / 1e12))
dt = t(lambda: b.copy_(a));   print("copy  (read+write): %.0f us  %.2f TB/s" % (dt * 1e6, 2 * n * 4 / dt / 1e12))
dt = t(lambda: a.sum());      print("sum   (read only) : %.0f us  %.2f TB/s" % (dt * 1e6, n * 4 / dt / 1e12))
