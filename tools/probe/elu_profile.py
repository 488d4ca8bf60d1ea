import sys; sys.path.insert(0, "/root/repo")
import dataclasses, collections, torch, kbnet_amd as kb
dev = torch.device("cuda:0")
cfg = dataclasses.replace(kb.kitti_config(), activation_func="elu")
m = kb.modules.KBNetModel.from_config(cfg, dev); m.load_state_dicts(*kb.synthetic.make_state_dicts(cfg, seed=0, gain=1.1))
fr = [f.to(dev) for f in kb.synthetic.make_frames(32, 352, 1216, "kitti", seed=1)]
m.forward(*fr); torch.cuda.synchronize()
kb.ops.PROFILE = []
m.forward(*fr); torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, work, ex, pipe, nb, s, e in kb.ops.PROFILE:
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += s.elapsed_time(e) * 1e3
kb.ops.PROFILE = None
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"{k:22s} {v[0]:3d} launches {v[1]:8.0f} us")
print("sum", round(tot), "us ->", round(32 / tot * 1e6), "frames/s eager")
