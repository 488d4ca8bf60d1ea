#!/usr/bin/env python3
"""Phase ablation of kbn_s2d_depth_front_forward (KBN_S2D_DEBUG bits: 1 vertical pass, 2 horizontal pass, 4 1x1 chain, 8 3x3 conv,
16 depth loads, 32 conv0_depth, 64 conv_depth + epilogue) at 32 KITTI frames, beside the two launches it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kbnet_amd as kb
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = kb.kitti_config()
m = kb.modules.KBNetModel.from_config(cfg, dev)
m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN["kitti"]))
fr = [f.to(dev) for f in kb.synthetic.make_frames(B, 352, 1216, "kitti", seed=1)]
x = torch.cat([fr[1], fr[2]], 1).contiguous()
enc, s2d = m.encoder, m.sparse_to_dense_pool
blk = enc.calibrated_backprojection1
c0d, cd = enc.conv0_depth, blk.conv_depth.conv_block[0]
kinv = kb.ops.intrinsics_inverse(fr[3])
packed_d = kb.ops.pack_kb1_depth_front_weight(c0d.conv.weight, cd.conv.weight, blk.proj_depth.conv.weight)
packed_s = kb.ops.pack_s2d_depth_front_weight([c.conv.weight for c in s2d.pool_convs], s2d.conv.conv.weight)
out = torch.empty((B, 16, 176, 608), device=dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


fused = lambda: kb.ops.s2d_depth_front(x, kinv, packed_s, packed_d, s2d.min_pool_sizes, s2d.max_pool_sizes, 16, 16, out, 0.2, 0.2, 0.2, 0.2)
print(f"fused: {timed(fused):.0f} us")
for bits in (1, 2, 4, 8, 16, 32, 64, 3, 12, 96, 127, 127 - 16):
    os.environ["KBN_S2D_DEBUG"] = str(bits)
    kb.ops.reload_env()
    print(f"  without bits {bits:3d}: {timed(fused):.0f} us")
os.environ.pop("KBN_S2D_DEBUG")
kb.ops.reload_env()
t1 = timed(lambda: s2d(x))
d = s2d(x)
t2 = timed(lambda: kb.ops.kb1_depth_front(d, kinv, packed_d, 16, 16, out, 0.2, 0.2, 0.2))
print(f"two launches: s2d {t1:.0f} + kb1_depth_front {t2:.0f} = {t1 + t2:.0f} us")
