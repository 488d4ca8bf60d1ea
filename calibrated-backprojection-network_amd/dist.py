"""Frame sharding across GPUs: one process per GPU, RCCL over xGMI.

Frames are independent (no normalisation layers, per-frame intrinsics), so the path
shards on the batch dimension with NO data-path collective; the only exchange is one
all-gather of the N/world x 1 x H x W depth maps per step (1.7 MB per KITTI frame).
This replaces the reference's three `torch.nn.DataParallel` wrappers
(reference src/kbnet_model.py:408-415), which scatter/replicate/gather per module
and per forward inside a single process.

Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU (tests).
"""

from __future__ import annotations

import os
from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process => 0, 0, 1)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of frame indices; the first n % world ranks get one extra."""
    q, r = divmod(n_frames, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_frames(tensors: Sequence[torch.Tensor], rank: int, world: int):
    lo, hi = shard_bounds(tensors[0].shape[0], rank, world)
    return [t[lo:hi] for t in tensors]


class ShardedRunner:
    """Runs `forward_fn` on this rank's frames and all-gathers the outputs.

    forward_fn(image, sparse_depth, validity, intrinsics) -> N_local x 1 x H x W.
    The gather buffer is allocated once and re-used; ranks may hold different frame
    counts (ragged tail) -- then the gather goes through padded slots."""

    def __init__(self, forward_fn: Callable, rank: int, world: int):
        self.forward_fn = forward_fn
        self.rank, self.world = rank, world
        self._buf = None
        self._ring = None      # pipelined mode: [(staging, gathered, work)] x 2
        self._step = 0

    def step_pipelined(self, local_inputs):
        """Forward of this step + an ASYNCHRONOUS all-gather of its outputs, overlapped with the
        next step's forward (RCCL runs on its own stream; xGMI traffic hides behind the MFMAs).
        Equal shard sizes only.  Returns the gathered N_total x 1 x H x W tensor of the PREVIOUS
        step (None on the first call); call `drain()` to get the last one."""
        out = self.forward_fn(*local_inputs)
        if self.world == 1:
            prev, self._last = getattr(self, "_last", None), out
            return prev
        if self._ring is None:
            mk = lambda: [torch.empty_like(out), torch.empty((self.world * out.shape[0],) + tuple(out.shape[1:]),
                                                             device=out.device, dtype=out.dtype), None]
            self._ring = [mk(), mk()]
        slot = self._ring[self._step & 1]
        if slot[2] is not None:
            slot[2].wait()                      # the gather that used this slot two steps ago
        slot[0].copy_(out)                      # `out` may be a graph's static buffer: detach it
        slot[2] = dist.all_gather_into_tensor(slot[1], slot[0], async_op=True)
        prev = self._ring[(self._step & 1) ^ 1]
        self._step += 1
        if prev[2] is not None:
            prev[2].wait()                      # orders the consumer after the previous gather
            return prev[1]
        return None

    def drain(self):
        if self.world == 1:
            return getattr(self, "_last", None)
        last = self._ring[(self._step - 1) & 1] if self._ring else None
        if last is None:
            return None
        if last[2] is not None:
            last[2].wait()
        return last[1]

    def step(self, local_inputs, n_total: Optional[int] = None, gather: bool = True):
        out = self.forward_fn(*local_inputs)
        if self.world == 1 or not gather:
            return out
        n_local = out.shape[0]
        n_total = n_total if n_total is not None else n_local * self.world
        per = -(-n_total // self.world)  # slot size (max frames on any rank)
        shape = (self.world * per,) + tuple(out.shape[1:])
        if self._buf is None or tuple(self._buf.shape) != shape or self._buf.device != out.device:
            self._buf = torch.empty(shape, device=out.device, dtype=out.dtype)
        if n_local == per:
            src = out.contiguous()
        else:
            src = torch.zeros((per,) + tuple(out.shape[1:]), device=out.device, dtype=out.dtype)
            src[:n_local] = out
        dist.all_gather_into_tensor(self._buf, src)
        if n_total == self.world * per:
            return self._buf
        parts = []
        for r in range(self.world):
            lo, hi = shard_bounds(n_total, r, self.world)
            parts.append(self._buf[r * per:r * per + (hi - lo)])
        return torch.cat(parts, dim=0)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
