import torch, faulthandler, sys
faulthandler.enable()
dev = torch.device("cuda:0")
x = torch.ones(1 << 20, device=dev)
outs = [torch.empty_like(x) for _ in range(4)]
s1 = torch.cuda.Stream(); a = torch.cuda.Stream(); b = torch.cuda.Stream()
torch.mul(x, 1.0, out=outs[0]); torch.cuda.synchronize()
mode = int(sys.argv[1])
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    if mode >= 4: b.wait_stream(cur)   # every stream forks from the capturing stream first; later edges are only waits
    if mode >= 1:
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            if mode >= 3:      # a fork inside the forked branch
                b.wait_stream(s1)
                with torch.cuda.stream(b):
                    torch.mul(x, 5.0, out=outs[3])
            torch.mul(x, 4.0, out=outs[2])
            if mode >= 3:
                s1.wait_stream(b)
    if mode >= 2:
        a.wait_stream(cur)
        with torch.cuda.stream(a):
            torch.mul(x, 3.0, out=outs[1])
    torch.mul(x, 2.0, out=outs[0])
    if mode >= 4: cur.wait_stream(b)
    if mode >= 2: cur.wait_stream(a)
    if mode >= 1: cur.wait_stream(s1)
g.replay(); torch.cuda.synchronize()
print("mode", mode, [float(o[0]) for o in outs])
