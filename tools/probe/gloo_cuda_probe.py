#!/usr/bin/env python3
"""Does the gloo backend move CUDA tensors (all_gather_into_tensor / all_reduce / barrier) between two processes that share ONE GPU?
If so, bench.py's whole N > 1 code path -- the real graphed forward on every rank -- can run at world size 2 on a one-GPU box
(KBN_BENCH_TEST_BACKEND=gloo-cuda), everything but RCCL itself."""
import os, socket, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        dist.init_process_group("gloo", rank=rank, world_size=2)
        dev = torch.device("cuda:0")
        x = torch.full((3, 4), float(rank + 1), device=dev)
        out = torch.empty(6, 4, device=dev)
        w = dist.all_gather_into_tensor(out, x, async_op=True)
        w.wait()
        t = torch.tensor([float(rank)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        q.put((rank, "ok", out[:, 0].tolist(), float(t)))
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, "error", f"{type(e).__name__}: {e}"[:500], None))

if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, port, q)) for r in range(2)]
    [p.start() for p in ps]
    for _ in ps:
        print(q.get(timeout=120))
    [p.join(10) for p in ps]
