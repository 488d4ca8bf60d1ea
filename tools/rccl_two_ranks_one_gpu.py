#!/usr/bin/env python3
"""Can TWO RCCL ranks share the ONE GPU of a leased box?  (VERDICT r4 next #6b.)

Spawns two processes that both use cuda:0, forms a 2-rank "nccl" (= RCCL) group and tries one all_gather_into_tensor of a
tiny tensor.  Prints what happened -- the collective's result, or the error text RCCL / torch raise -- as one line per rank
and a final verdict; never hangs longer than --timeout seconds.  DESIGN.md section 5 quotes the outcome.

    python tools/rccl_two_ranks_one_gpu.py [--timeout 120]
"""
import argparse
import os
import socket
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp


def worker(rank, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import datetime
    import torch.distributed as dist
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=60))
        x = torch.full((4,), float(rank + 1), device="cuda:0")
        out = torch.empty(8, device="cuda:0")
        dist.all_gather_into_tensor(out, x)
        torch.cuda.synchronize()
        q.put((rank, "ok", out.cpu().tolist()))
        dist.destroy_process_group()
    except Exception as e:   # the error text is the result
        q.put((rank, "error", f"{type(e).__name__}: {e}".strip().replace("\n", " | ")[:1500]))
        traceback.print_exc()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--timeout", type=int, default=120)
    args = ap.parse_args()
    print(f"devices visible: {torch.cuda.device_count()}; torch {torch.__version__}; RCCL {'.'.join(map(str, torch.cuda.nccl.version()))}")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = []
    try:
        for _ in procs:
            results.append(q.get(timeout=args.timeout))
    except Exception:
        results.append((-1, "timeout", f"no answer within {args.timeout} s (a rank hangs inside group creation or the collective)"))
    for p in procs:
        p.join(timeout=10)
        if p.is_alive():
            p.kill()
    for r in sorted(results):
        print("rank %d: %s: %s" % r)
    ok = len(results) == 2 and all(r[1] == "ok" for r in results)
    print("VERDICT: two RCCL ranks on one device " + ("WORK" if ok else "do NOT work on this box"))


if __name__ == "__main__":
    main()
