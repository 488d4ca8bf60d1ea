import sys
sys.path.insert(0, ".")
import torch, kbnet_amd as kb
dev = torch.device("cuda:0")
for shape in [(32, 3, 352, 1216), (32, 48, 176, 608), (32, 8, 352, 1216)]:
    x = torch.randn(*shape, device=dev)
    st = kb.ops.ActStats(shape[0], dev)
    slot = st.new()
    f = lambda: kb.ops.absmax_frames(x, slot)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 50
    s.record()
    for _ in range(20): y = x.abs().amax(dim=(1, 2, 3))
    e.record(); torch.cuda.synchronize()
    print(shape, f"absmax_frames {us:.1f} us = {x.numel()*4/us/1e6:.2f} TB/s; torch abs().amax {s.elapsed_time(e)*50:.1f} us")
