"""CPU-baseline calibration: seconds per KITTI frame of the oracle (the CPU restatement of the reference) for a range of
torch thread counts -- how bench.py's cpu_baseline chose 16 threads.  Lives under tests/ because it runs the oracle."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kbnet_amd as kb
from oracle import kbnet_oracle as orc
cfg = kb.kitti_config(); sds = kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN["kitti"])
frames = kb.synthetic.make_frames(1, 352, 1216, "kitti", seed=1)
print("cpu_count", os.cpu_count())
for t in (8, 16, 32, 64, 128):
    torch.set_num_threads(t)
    run = lambda: orc.kbnet_forward(*frames, *sds, cfg.min_pools, cfg.max_pools, 1.5, 100.0)
    run(); t0 = time.perf_counter(); run(); run(); dt = (time.perf_counter() - t0) / 2
    print(t, "threads:", round(dt, 3), "s/frame", flush=True)
