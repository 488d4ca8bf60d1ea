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
    Gather buffers are allocated once per output shape and re-used; ranks may hold different
    frame counts (ragged tail) -- then the gather goes through padded slots.

    Lifetime of returned tensors: they are views of the runner's own buffers.  `step()` / `step_mixed()`
    results are overwritten by the next call with the same output shape; `step_pipelined()` results by the
    next-but-one call when a collective runs (they live in the runner's two gather buffers) -- and by the NEXT call
    without one (a single rank: the result is then the forward's own output tensor, which with rotating graph outputs
    is the tensor the next replay writes).  Clone what must live longer."""

    def __init__(self, forward_fn: Callable, rank: int, world: int, gather_single: bool = False):
        """`gather_single`: run the collectives even with one rank (a one-rank RCCL group is legal): lets a single-GPU
        box exercise the whole RCCL path -- stream ordering against graph replays included -- in tests."""
        self.forward_fn = forward_fn
        self.rank, self.world = rank, world
        self.collective = world > 1 or (gather_single and dist.is_initialized())
        self._bufs = {}        # (per, shape[1:], device) -> gather buffer
        self._ring = None      # pipelined mode: [(staging, gathered, work)] x 2
        self._step = 0

    # ---- fixed-shape stream, gather of step i overlapped with the forward of step i+1 -------------------
    def step_pipelined(self, local_inputs):
        """Forward of this step + an ASYNCHRONOUS all-gather of its outputs, overlapped with the
        next step's forward (RCCL runs on its own stream; xGMI traffic hides behind the MFMAs).
        Every rank must hold the same number of frames (checked once).  Returns the gathered
        N_total x 1 x H x W tensor of the PREVIOUS step (None on the first call); `drain()` returns the last."""
        # A forward with `rotating_outputs` >= 2 (GraphedForward(outputs=2): the captured graph exists once per output tensor and
        # calls alternate between them) hands back a tensor nothing overwrites before the next-but-one call: the all-gather
        # reads it in place and the ring's staging copy (55 MB per rank and step at 32 KITTI frames) is skipped.  The gather
        # that read the tensor this call is about to overwrite was issued two steps ago; it is awaited here, BEFORE the forward.
        rotating = int(getattr(self.forward_fn, "rotating_outputs", 0) or 0) >= 2
        if rotating and self._ring is not None:
            # by tensor identity, not by step parity: somebody else may have called the forward in between (bench.py's gather check
            # does), which shifts the rotation against this runner's step count
            nxt = self.forward_fn.next_output() if hasattr(self.forward_fn, "next_output") else None
            for stale in self._ring:
                if stale[2] is not None and (nxt is None or stale[0] is None or stale[0].data_ptr() == nxt.data_ptr()):
                    stale[2].wait()
                    stale[2] = None
        out = self.forward_fn(*local_inputs)
        if self._ring is None:
            if self.collective:   # a ragged shard would hang or fail inside the collective: check once
                n = torch.tensor([out.shape[0], -out.shape[0]], device=out.device, dtype=torch.int64)
                dist.all_reduce(n, op=dist.ReduceOp.MAX)
                if int(n[0]) != -int(n[1]):
                    raise ValueError(f"step_pipelined needs equal shards on every rank (min {-int(n[1])}, "
                                     f"max {int(n[0])} frames); use step() / step_mixed() for ragged batches")
            gathered = lambda: (torch.empty((self.world * out.shape[0],) + tuple(out.shape[1:]),
                                            device=out.device, dtype=out.dtype) if self.collective else None)
            self._ring = [[None if rotating else torch.empty_like(out), gathered(), None] for _ in range(2)]
        slot = self._ring[self._step & 1]
        if slot[2] is not None:
            slot[2].wait()                      # the gather that used this slot two steps ago
        if rotating:
            slot[0] = out                       # the forward's own (rotating) output tensor: no copy
        else:
            slot[0].copy_(out)                  # `out` may be a graph's ONE static buffer: detach it
        if self.collective:
            slot[2] = dist.all_gather_into_tensor(slot[1], slot[0], async_op=True)
        prev = self._ring[(self._step & 1) ^ 1]
        self._step += 1
        if self._step == 1:
            return None
        return self._finish(prev)

    def _finish(self, slot):
        if not self.collective:
            return slot[0]
        if slot[2] is not None:
            slot[2].wait()                      # orders the consumer after the gather
        return slot[1]

    def drain(self):
        if self._ring is None or self._step == 0:
            return None
        return self._finish(self._ring[(self._step - 1) & 1])

    # ---- one batch, any split ----------------------------------------------------------------------------
    def _slots(self, out, per, bucket=0):
        # `bucket`: several gathers of one step_mixed call are in flight at once, so two buckets of the same frame
        # shape and slot size must not share a buffer (the second collective would overwrite the first)
        key = (per, tuple(out.shape[1:]), out.device, out.dtype, bucket)
        buf = self._bufs.get(key)
        if buf is None:
            buf = self._bufs[key] = torch.empty((self.world * per,) + tuple(out.shape[1:]), device=out.device,
                                                dtype=out.dtype)
        return buf

    def _gather_start(self, out, n_total, bucket=0):
        """-> (buffer, work, per): `out` (N_local frames, N_local possibly 0 < per) into padded slots."""
        per = -(-n_total // self.world)          # slot size (max frames on any rank)
        buf = self._slots(out, per, bucket)
        if out.shape[0] == per:
            src = out.contiguous()
        else:
            src = torch.zeros((per,) + tuple(out.shape[1:]), device=out.device, dtype=out.dtype)
            src[:out.shape[0]] = out
        return buf, dist.all_gather_into_tensor(buf, src, async_op=True), per

    def _gather_finish(self, buf, work, per, n_total):
        work.wait()
        if n_total == self.world * per:
            return buf
        parts = []
        for r in range(self.world):
            lo, hi = shard_bounds(n_total, r, self.world)
            parts.append(buf[r * per:r * per + (hi - lo)])
        return torch.cat(parts, dim=0)

    def step(self, local_inputs, n_total: Optional[int] = None, gather: bool = True):
        out = self.forward_fn(*local_inputs)
        if not self.collective or not gather:
            return out
        n_total = n_total if n_total is not None else out.shape[0] * self.world
        lo, hi = shard_bounds(n_total, self.rank, self.world)
        if out.shape[0] != hi - lo:
            raise ValueError(f"rank {self.rank} holds {out.shape[0]} frames, shard_bounds gives {hi - lo} of {n_total}")
        return self._gather_finish(*self._gather_start(out, n_total), n_total)

    # ---- mixed-shape stream (BASELINE config 5) ----------------------------------------------------------
    def step_mixed(self, buckets: Sequence[Tuple[Callable, Optional[Sequence[torch.Tensor]], int, Tuple[int, ...]]]):
        """One step of a stream that mixes frame shapes (VOID 480x640 / NYUv2 416x576 / KITTI 352x1216, each
        with its own weights and per-frame intrinsics).  `buckets`: one entry per shape, in the SAME order on
        every rank: (forward_fn, this rank's inputs of that shape or None, n_total frames of that shape over all
        ranks, output shape per frame e.g. (1, H, W)).  Frames of a bucket are split with `shard_bounds`, so a
        rank may hold fewer frames than its neighbours or none (it then contributes an empty, padded slot).
        One all-gather per shape bucket into that bucket's own buffer; the gather of bucket i is in flight
        while bucket i+1 computes.  Returns the list of N_total x ... tensors, bucket order."""
        pending = []
        for bucket, (forward_fn, inputs, n_total, frame_shape) in enumerate(buckets):
            lo, hi = shard_bounds(n_total, self.rank, self.world)
            out = None
            if inputs is not None and hi > lo:
                out = forward_fn(*inputs)
                if out.shape[0] != hi - lo or tuple(out.shape[1:]) != tuple(frame_shape):
                    raise ValueError(f"bucket {tuple(frame_shape)}: forward returned {tuple(out.shape)}, this rank's "
                                     f"share is {hi - lo} frames")
            elif hi > lo:
                raise ValueError(f"rank {self.rank} owns {hi - lo} frames of bucket {tuple(frame_shape)} but got no inputs")
            if not self.collective:
                pending.append((out, None, 0, n_total))
                continue
            if out is None:   # nothing of this shape here: still take part in the collective
                ref = inputs[0] if inputs is not None else None
                dev = ref.device if ref is not None else self._default_device()
                out = torch.empty((0,) + tuple(frame_shape), device=dev, dtype=torch.float32)
            pending.append(self._gather_start(out, n_total, bucket) + (n_total,))
        results = []
        for buf, work, per, n_total in pending:
            results.append(buf if work is None else self._gather_finish(buf, work, per, n_total))
        return results

    def _default_device(self):
        if dist.is_initialized() and dist.get_backend() == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value: float, device) -> list:
    """`value` of every rank, rank order (a one-element list without a process group): per-rank rates of a run."""
    if not dist.is_initialized():
        return [float(value)]
    world = dist.get_world_size()
    t = torch.zeros(world, dtype=torch.float64, device=device)
    t[dist.get_rank()] = value
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def sum_over_ranks(values: torch.Tensor) -> torch.Tensor:
    """Element-wise sum of a small tensor over all ranks (in place; identity without a process group)."""
    if dist.is_initialized():
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
    return values


def mean_metrics_over_ranks(per_frame: torch.Tensor) -> torch.Tensor:
    """The reference's evaluation summary over a sharded run: it averages the per-sample MAE / RMSE / iMAE / iRMSE over
    ALL samples (reference src/kbnet.py:952-984, np.mean over the arrays filled at :932-950).  `per_frame`: this
    rank's N_local x 4 metrics from ops.eval_metrics (N_local may be 0 and may differ between ranks); returns the 4
    means over the frames of every rank (fp64, on per_frame's device).  ONE all-reduce of five numbers."""
    acc = torch.zeros(5, dtype=torch.float64, device=per_frame.device)
    if per_frame.numel():
        acc[:4] = per_frame.to(torch.float64).sum(dim=0)
    acc[4] = per_frame.shape[0]
    sum_over_ranks(acc)
    return acc[:4] / acc[4].clamp_min(1.0)
