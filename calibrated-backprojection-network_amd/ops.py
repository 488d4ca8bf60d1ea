"""Tensor-level wrappers over the C ABI.  PyTorch is plumbing here: it owns the
device memory and the stream; every computation below is a HIP kernel launch
through libkbnet_hip.so.  CPU tensors are rejected (no fallback).
"""

from __future__ import annotations

import contextlib
import ctypes as C
import functools
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import ConvSrc, KbnError, check


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _on_tensor_device(fn):
    """The C side launches on the calling thread's CURRENT device and on the stream it is handed, so every
    wrapper below runs with the tensors' device current (reference modules accept any device:
    `KBNetModel(..., device=cuda:1)` while cuda:0 is current, DataParallel replicas on per-device
    threads) and takes torch's current stream of THAT device.  Tensors on different devices are an error."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        idx = None
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if idx is None:
                    idx = a.device.index
                elif a.device.index != idx:
                    raise KbnError(f"{fn.__name__}: tensors live on different devices (cuda:{idx} and {a.device})")
        if idx is None or idx == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(idx):
            return fn(*args, **kwargs)
    return wrapper


def reload_env():
    """Re-reads the KBN_* switches from the environment (the library reads them once at load time).  The host mirror's
    own A/B switches (KBN_NO_PAIR*, KBN_NO_OVERLAP, ...) are answered by the library too (`knob`), so they follow."""
    _lib.load().kbn_reload_env()


def knob(name: str) -> int:
    """Value of a KBN_* switch as the library read it (kbn_knob): 0 when unset."""
    return int(_lib.load().kbn_knob(name.encode()))


def split_enabled() -> bool:
    """False under KBN_NO_SPLIT=1: every split-operand entry point declines (A/B against the all-fp32-MFMA path)."""
    return knob("KBN_NO_SPLIT") == 0


@contextlib.contextmanager
def autotune(enabled: bool = True):
    """First-use tuning of launch geometry is OFF by default (ABI calls only enqueue work).  Inside this
    context the first launch of every problem shape times its candidate geometries on the stream and
    waits for its own events (kbn_set_autotune); the choices stay cached for the process."""
    lib = _lib.load()
    old = lib.kbn_set_autotune(1 if enabled else 0)
    try:
        yield
    finally:
        lib.kbn_set_autotune(old)


# Optional per-launch timing (bench.py): when PROFILE is a list, every ABI call is
# bracketed by HIP events recorded on the launch stream and
# (name, work, executed, pipe, nbytes, start, end) is appended.  `work` is the launch's ALGORITHMIC FLOPs
# (the reference's direct formulation: convs, and the conv part of S2D / the head; 0 for pure byte movers); `executed` the FLOPs the launch really
# issues on the matrix cores (Winograd / phase-decomposed up-convs execute fewer; tile and channel padding
# execute more), None for byte-bound launches; `pipe` names the pipe those MFMAs run on -- "fp32"
# (v_mfma_f32_*_f32: the vector datapath's 157.3 TFLOP/s), "fp16" (split-operand kernels: three fp16 MFMAs per fp32
# product on the 2.5 PFLOP/s matrix core) -- so that whoever prices `executed`
# uses the right peak; `nbytes` the launch's ALGORITHMIC HBM bytes (every input read once + every output written once).
PROFILE = None
PIPE_PEAK_TFLOPS = {"fp32": 157.3, "fp16": 2500.0}    # MI355X_MICROARCH.md, dense
PIPE_PRODUCTS = {"fp32": 1.0, "fp16": 3.0}    # MFMA products issued per product of the reference


def _launch(name: str, work: float, fn, executed=None, pipe: Optional[str] = None, nbytes: Optional[float] = None):
    if PROFILE is None:
        return fn()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    status = fn()
    end.record()
    PROFILE.append((name, work, executed() if callable(executed) else executed, pipe, nbytes, start, end))
    return status


def _src_bytes(srcs, n: int) -> float:
    """fp32-equivalent bytes the tensor / pair sources of a conv hold (4 B per value in both formats)."""
    total = 0
    for s in srcs:
        if s.kind in (_lib.KBN_SRC_TENSOR, _lib.KBN_SRC_PAIR):
            total += s.channels * s.src_height * s.src_width
    return 4.0 * n * total


def conv_executed_flops(n, out_channels, in_channels, kernel_size, stride, in_height, in_width, resize=False):
    """FLOPs kbn_conv2d_forward issues on the matrix cores for this problem, padding included, from the
    launch plan (kbn_conv2d_query): Winograd F(2x2,3x3) regions of 64 tiles x 16 frequencies x 64-channel
    output tiles; direct kernels 4*MW m-blocks of 16 pixels x NB n-blocks of 16 channels x padded K."""
    pl = conv_plan(n, out_channels, in_channels, kernel_size, stride, in_height, in_width, resize)
    if pl["kernel"] == "wino":
        return 2.0 * pl["workgroups"] * 64 * 16 * in_channels * 64
    cpad = -(-in_channels // pl["CK"]) * pl["CK"]
    return 2.0 * pl["workgroups"] * (4 * pl["MW"] * 16) * (pl["NB"] * 16) * cpad * kernel_size * kernel_size


def upconv2x_executed_flops(n, in_channels, out_channels, src_height, src_width, plain=False):
    """`plain`: the four-phase form whatever the shape (the transposed conv: kbn_deconv2x_forward)."""
    info = (C.c_int * 4)()
    check(_lib.load().kbn_upconv2x_query(n, in_channels, out_channels, src_height, src_width, info), "kbn_upconv2x_query")
    if plain and info[0] == 9:   # the 9-product form has its own filter tiling: the four-phase one pads to make_up2x_plan's tiles
        nblk = -(-out_channels // 16)
        best, bestpad = 1, 1 << 30
        for nb in range(1, 5):
            pad = -(-nblk // nb) * nb
            if pad < bestpad or (pad == bestpad and nb > best):
                best, bestpad = nb, pad
        info[1] = bestpad * 16
    return 2.0 * n * src_height * src_width * (16 if plain else info[0]) * info[1] * info[2]


_PLAN_CACHE = {}


def conv_plan(n: int, out_channels: int, in_channels: int, kernel_size: int, stride: int, in_height: int,
              in_width: int, resize: bool = False):
    """Kernel variant the library picks (kbn_conv2d_query): dict with CK, NB, MW, TWB, TH,
    workgroups, maxpos, pipelined (3 = conv_wino_kernel, where MW x TWB is the region's tile
    rows x columns; 2 = conv_dma_kernel; 1/0 = conv_igemm_kernel) and `kernel` (its name)."""
    key = (n, out_channels, in_channels, kernel_size, stride, in_height, in_width, int(resize))
    if key not in _PLAN_CACHE:
        info = (C.c_int * 8)()
        check(_lib.load().kbn_conv2d_query(*key, info), "kbn_conv2d_query")
        d = dict(zip(("CK", "NB", "MW", "TWB", "TH", "workgroups", "maxpos", "pipelined"), info))
        d["kernel"] = {3: "wino", 2: "dma"}.get(d["pipelined"], "igemm")
        _PLAN_CACHE[key] = d
    return _PLAN_CACHE[key]


def _require(t: torch.Tensor, name: str, ndim: Optional[int] = None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise KbnError(f"{name}: expected a CUDA/HIP tensor (the HIP path has no CPU fallback)")
    if t.dtype != torch.float32:
        raise KbnError(f"{name}: expected float32, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise KbnError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")


def _planes(t: torch.Tensor, name: str):
    """(ptr, batch_stride) of an N x C x H x W tensor whose frames are dense C x H x W blocks
    (a contiguous tensor or a channel slice of one)."""
    _require(t, name, 4)
    n, c, h, w = t.shape
    if t.stride(3) != 1 or t.stride(2) != w or t.stride(1) != h * w:
        raise KbnError(f"{name}: planes must be dense (got strides {t.stride()})")
    return t.data_ptr(), (t.stride(0) if n > 1 else c * h * w)


def _int_array(values: Sequence[int]):
    arr = (C.c_int * max(len(values), 1))(*values)
    return arr


# ------------------------------------------------------------- activation statistics
class ActStats:
    """Per-frame max |a| slots of the tensors of one forward (include/kbnet_hip.h, "activation statistics"): one zeroed
    int32 buffer [capacity][n]; `new()` hands out a row.  A conv launch folds max |out| of every frame into the slot
    it is given (`out_absmax=`); `tensor_src(t, absmax=slot)` hands it to the consumer, whose split-operand kernel
    derives its fp16 window from it on the device, frame by frame.  Allocated inside the forward (also under graph
    capture: the zero fill is a node of the graph), so nothing outlives the call."""

    def __init__(self, n: int, device, capacity: int = 40):
        self.buf = torch.zeros((capacity, n), device=device, dtype=torch.int32)
        self.n, self.used = n, 0
        self.unfilled = set()   # data_ptr of slots handed to a launch that does not fill them (a layer-by-layer activation pass in between)

    def skip(self, slot: Optional[torch.Tensor]):
        """A producer that writes (part of) the slot's tensor without folding its maxima: consumers measure the tensor."""
        if slot is not None:
            self.unfilled.add(slot.data_ptr())

    def usable(self, slot_ptr) -> bool:
        return bool(slot_ptr) and slot_ptr not in self.unfilled

    def new(self) -> torch.Tensor:
        if self.used == self.buf.shape[0]:   # more tensors than foreseen: chain another buffer (one more fill launch)
            self.buf = torch.zeros_like(self.buf)
            self.used = 0
        slot = self.buf[self.used]
        self.used += 1
        return slot

    def measure(self, t: torch.Tensor) -> torch.Tensor:
        """A slot filled by a pass over `t` (kbn_absmax_frames): tensors that come from outside the forward."""
        return absmax_frames(t, self.new())


def _slot_ptr(slot: Optional[torch.Tensor], n: int):
    if slot is None:
        return None
    if slot.dtype != torch.int32 or slot.dim() != 1 or slot.shape[0] != n or not slot.is_contiguous() or not slot.is_cuda:
        raise KbnError(f"absmax slot: expected a contiguous int32 device tensor of {n} elements")
    return slot.data_ptr()


@_on_tensor_device
def absmax_frames(t: torch.Tensor, slot: torch.Tensor) -> torch.Tensor:
    """Folds max |t[i]| of every frame i into slot[i] (bit patterns; kbn_absmax_frames).  No synchronisation."""
    ptr, bs = _planes(t, "t")
    n, c, h, w = t.shape
    check(_launch("absmax", 4.0 * n * c * h * w,
                  lambda: _lib.load().kbn_absmax_frames(ptr, bs, n, c * h * w, _slot_ptr(slot, n), _stream()),
                  nbytes=4.0 * n * c * h * w), "kbn_absmax_frames")
    return slot


def slot_values(slot: torch.Tensor) -> torch.Tensor:
    """The per-frame maxima a slot holds, as floats (diagnostics, tests; synchronises when read)."""
    return slot.view(torch.float32)


# ----------------------------------------------------------------------------- S2D
@_on_tensor_device
def s2d_forward(x, w_pool_convs: List[torch.Tensor], w_conv, min_pool_sizes, max_pool_sizes,
                negative_slope: float = 0.2, out: Optional[torch.Tensor] = None):
    lib = _lib.load()
    _require(x, "x", 4)
    x = x.contiguous()
    n, cin, h, w = x.shape
    ws = [wt.detach().contiguous() for wt in w_pool_convs]
    for i, wt in enumerate(ws):
        _require(wt, f"pool_convs.{i}.weight", 4)
    wc = w_conv.detach().contiguous()
    _require(wc, "conv.weight", 4)
    nf = wc.shape[0]
    mins = [int(s) for s in min_pool_sizes if s > 1]
    maxs = [int(s) for s in max_pool_sizes if s > 1]
    if ws[0].shape[1] != len(mins) + len(maxs) or wc.shape[1] != nf + cin:
        raise KbnError("S2D weight shapes do not match the pool lists / input channels")
    if out is None:
        out = torch.empty((n, nf, h, w), device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (n, nf, h, w) or not out.is_contiguous():
        raise KbnError("s2d_forward: `out` must be a contiguous N x n_filter x H x W tensor")
    wptrs = (C.c_void_p * len(ws))(*[wt.data_ptr() for wt in ws])
    amin, amax = _int_array(mins), _int_array(maxs)
    # (work = the layer's multiply-adds x 2: nothing is skipped or padded, so executed = algorithmic; its bytes travel as `nbytes`)
    check(_launch("s2d", 2.0 * n * h * w * (ws[0].shape[1] * nf + (len(ws) - 1) * nf * nf + 9 * (nf + cin) * nf),
                  lambda: lib.kbn_s2d_forward(x.data_ptr(), wptrs, wc.data_ptr(), out.data_ptr(), n, h, w, cin,
                                              amin, len(mins), amax, len(maxs), len(ws), nf,
                                              float(negative_slope), _stream()),
                  # both convs on v_mfma_f32_4x4x1 (the fp32 datapath): 2 x (pools x nf + (convs - 1) x nf^2 + 9 (nf + cin) nf) per pixel
                  executed=2.0 * n * h * w * (ws[0].shape[1] * nf + (len(ws) - 1) * nf * nf + 9 * (nf + cin) * nf),
                  pipe="fp32", nbytes=4.0 * n * h * w * (cin + nf)), "kbn_s2d_forward")
    return out


@_on_tensor_device
def s2d_pyramid(x, min_pool_sizes, max_pool_sizes):
    lib = _lib.load()
    _require(x, "x", 4)
    x = x.contiguous()
    n, cin, h, w = x.shape
    mins = [int(s) for s in min_pool_sizes if s > 1]
    maxs = [int(s) for s in max_pool_sizes if s > 1]
    out = torch.empty((n, len(mins) + len(maxs), h, w), device=x.device, dtype=torch.float32)
    check(lib.kbn_s2d_pyramid(x.data_ptr(), cin * h * w, out.data_ptr(), n, h, w, _int_array(mins),
                              len(mins), _int_array(maxs), len(maxs), _stream()), "kbn_s2d_pyramid")
    return out


# ---------------------------------------------------------------------- intrinsics
@_on_tensor_device
def intrinsics_inverse(intrinsics, scale_x: float = 1.0, scale_y: float = 1.0):
    lib = _lib.load()
    _require(intrinsics, "intrinsics", 3)
    k = intrinsics.contiguous()
    if k.shape[1:] != (3, 3):
        raise KbnError("intrinsics must be N x 3 x 3")
    out = torch.empty_like(k)
    check(lib.kbn_intrinsics_inverse(k.data_ptr(), out.data_ptr(), k.shape[0], float(scale_x),
                                     float(scale_y), _stream()), "kbn_intrinsics_inverse")
    return out


@_on_tensor_device
def camera_coordinates(kinv, height: int, width: int):
    lib = _lib.load()
    _require(kinv, "kinv", 3)
    kinv = kinv.contiguous()
    out = torch.empty((kinv.shape[0], 3, height, width), device=kinv.device, dtype=torch.float32)
    check(lib.kbn_camera_coordinates(kinv.data_ptr(), out.data_ptr(), kinv.shape[0], height, width,
                                     _stream()), "kbn_camera_coordinates")
    return out


# --------------------------------------------------------------------------- conv2d
def _reusable(blob, nfloats, like):
    return (blob is not None and blob.numel() == nfloats and blob.device == like.device and
            blob.dtype == torch.float32 and blob.is_contiguous())


@_on_tensor_device
def pack_conv_weight(weight: torch.Tensor, stride: int = 1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """OIHW -> MFMA fragment order for a conv of this stride (done once per weight; see
    _PackedWeight in modules.py).  `out`: an existing blob of the right size to re-pack into (keeps
    the device pointer a captured graph holds valid)."""
    lib = _lib.load()
    w = weight.detach().contiguous()
    _require(w, "weight", 4)
    oc, cin, kh, kw = w.shape
    if kh != kw:
        raise KbnError("square kernels only")
    nbytes = lib.kbn_conv2d_packed_weight_bytes(oc, cin, kh, stride)
    if nbytes == 0:
        raise KbnError(f"unsupported conv weight shape {tuple(w.shape)}")
    packed = out if _reusable(out, nbytes // 4, w) else torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    check(lib.kbn_conv2d_pack_weight(w.data_ptr(), packed.data_ptr(), oc, cin, kh, stride, _stream()),
          "kbn_conv2d_pack_weight")
    return packed


def tensor_src(t: torch.Tensor, name="src", absmax: Optional[torch.Tensor] = None) -> ConvSrc:
    """`absmax`: the tensor's per-frame max |a| slot (ActStats), when its producer filled one."""
    ptr, bs = _planes(t, name)
    s = ConvSrc()
    s.kind = _lib.KBN_SRC_TENSOR
    s.channels = t.shape[1]
    s.data = ptr
    s.batch_stride = bs
    s.src_height, s.src_width = t.shape[2], t.shape[3]
    s.absmax = _slot_ptr(absmax, t.shape[0])
    s._keep = (t, absmax)   # the struct holds raw pointers
    return s


class PairTensor:
    """An activation tensor in the producer-written split format (include/kbnet_hip.h, "PAIR tensors"): per frame
    [channels / 8][h1 | h2][H * W + 1 pixels][8 channels] fp16 with a 2^k = h1 + 2^-11 h2, the per-frame 2^k in `scale`
    (written by the producing kernel) and the true max |a| per frame in the `absmax` slot.  Only tensors that split-operand
    kernels alone read are kept this way: the decoder's concat-conv outputs (conv3x3_split(out=PairTensor), read by the
    next up-conv through pair_src).  Same bytes as fp32; the consumers stage it by LDS-DMA instead of splitting it."""

    LOG = None   # diagnostics (tests/analysis): when a list hangs here, every PairTensor of a forward is appended

    def __init__(self, n: int, channels: int, height: int, width: int, device, stats: "ActStats"):
        if channels % 8:
            raise KbnError("PairTensor: channels must be a multiple of 8")
        if PairTensor.LOG is not None:
            PairTensor.LOG.append(self)
        self.shape = (n, channels, height, width)
        self.data = torch.empty((n, channels // 8, 2, height * width + 1, 8), device=device, dtype=torch.float16)
        self.scale = torch.empty(n, device=device, dtype=torch.float32)
        self.absmax = stats.new()
        # stride-2 producers (conv3x3_split(stride=2, out=PairTensor)) can also write the pixels (2y, 2x) in fp32, as a dense
        # N x C x ceil(H/2) x ceil(W/2) tensor: what the next KB level's 1x1 stride-2 conv_fused reads of this tensor
        self.sub: Optional[torch.Tensor] = None

    @property
    def device(self):
        return self.data.device

    def with_sub(self) -> "PairTensor":
        n, c, h, w = self.shape
        self.sub = torch.empty((n, c, (h + 1) // 2, (w + 1) // 2), device=self.data.device, dtype=torch.float32)
        return self

    def window_slack_log2(self) -> torch.Tensor:
        """Per frame: binades between the top of the fp16 window the producer chose from its BOUND (2^15 / scale) and the true
        max |a| it then measured (the absmax slot).  The two terms keep 22 bits down to 2^-29 of the window, so a slack
        below ~16 costs nothing (tests/test_split_math_cpu.py); diagnostics only (synchronises)."""
        amax = slot_values(self.absmax).double().clamp_min(1e-300)
        return 15.0 - torch.log2(amax * self.scale.double())

    def float(self) -> torch.Tensor:
        """The tensor as N x C x H x W fp32 (tests, diagnostics)."""
        n, c, h, w = self.shape
        v = self.data[:, :, 0, :h * w].double() + self.data[:, :, 1, :h * w].double() / 2048.0   # n, c/8, hw, 8
        v = v / self.scale.double().view(n, 1, 1, 1)
        return v.permute(0, 1, 3, 2).reshape(n, c, h, w).float()


def pair_src(t: PairTensor, name="src") -> ConvSrc:
    n, c, h, w = t.shape
    s = ConvSrc()
    s.kind = _lib.KBN_SRC_PAIR
    s.channels = c
    s.data = t.data.data_ptr()
    s.batch_stride = t.data.stride(0)
    s.src_height, s.src_width = h, w
    s.absmax = _slot_ptr(t.absmax, n)
    s.scale = t.scale.data_ptr()
    s._keep = (t,)
    return s


def coords_src(kinv: torch.Tensor) -> ConvSrc:
    _require(kinv, "kinv", 3)
    s = ConvSrc()
    s.kind = _lib.KBN_SRC_COORDS
    s.channels = 3
    s.kinv = kinv.data_ptr()
    s._keep = (kinv,)   # the struct holds raw pointers: the tensors live as long as it does
    return s


def xyz_src(depth: torch.Tensor, proj_weight: torch.Tensor, kinv: Optional[torch.Tensor], coordinates: Optional[torch.Tensor] = None) -> ConvSrc:
    """The KB layer's backprojection channels coords * act(proj . depth), computed in-kernel; coords = K^-1 [x y 1]^T from `kinv`
    (N x 3 x 3) or read from the dense `coordinates` tensor (N x 3 x H x W, the reference's own argument)."""
    ptr, bs = _planes(depth, "depth")
    s = ConvSrc()
    s.kind = _lib.KBN_SRC_XYZ
    s.channels = 3
    s.data = ptr
    s.batch_stride = bs
    s.aux_channels = depth.shape[1]
    s.proj_weight = proj_weight.data_ptr()
    if coordinates is not None:
        _require(coordinates, "coordinates", 4)
        if not coordinates.is_contiguous() or tuple(coordinates.shape) != (depth.shape[0], 3, depth.shape[2], depth.shape[3]):
            raise KbnError("coordinates must be a contiguous N x 3 x H x W tensor of the depth features' size")
        s.coordinates = coordinates.data_ptr()
        s.coordinates_batch_stride = coordinates.stride(0)
    else:
        _require(kinv, "kinv", 3)
        s.kinv = kinv.data_ptr()
    s._keep = (depth, proj_weight, kinv, coordinates)
    return s


@_on_tensor_device
def conv2d(srcs: List[ConvSrc], packed_weight: torch.Tensor, n: int, out_channels: int,
           kernel_size: int, stride: int, in_height: int, in_width: int, out: torch.Tensor,
           resize: bool = False, negative_slope: Optional[float] = 0.2, out_absmax: Optional[torch.Tensor] = None):
    """`out` is an N x out_channels x ceil(H/s) x ceil(W/s) tensor or channel slice; `out_absmax`: slot that receives
    max |out| per frame."""
    lib = _lib.load()
    arr = (ConvSrc * len(srcs))(*srcs)
    optr, obs = _planes(out, "out")
    oh, ow = -(-in_height // stride), -(-in_width // stride)
    if tuple(out.shape) != (n, out_channels, oh, ow):
        raise KbnError(f"out has shape {tuple(out.shape)}, expected {(n, out_channels, oh, ow)}")
    cin = sum(s.channels for s in srcs)
    name = "conv_igemm"
    if PROFILE is not None:
        pl = conv_plan(n, out_channels, cin, kernel_size, stride, in_height, in_width, resize)
        if pl["kernel"] == "wino":
            name = "conv_wino"
        else:
            name = ("conv_dma" if pl["pipelined"] == 2 else "conv_igemm") + \
                f"<{kernel_size},{stride},{pl['CK']},{pl['NB']},{pl['MW']}>"
    check(_launch(name,
                  2.0 * n * oh * ow * cin * kernel_size * kernel_size * out_channels,
                  lambda: lib.kbn_conv2d_forward(arr, len(srcs), packed_weight.data_ptr(), optr, obs, n,
                                                 out_channels, kernel_size, stride, in_height, in_width,
                                                 _lib.KBN_RESIZE_NEAREST if resize else _lib.KBN_RESIZE_NONE,
                                                 0 if negative_slope is None else 1,
                                                 0.0 if negative_slope is None else float(negative_slope),
                                                 _slot_ptr(out_absmax, n), _stream()),
                  executed=lambda: conv_executed_flops(n, out_channels, cin, kernel_size, stride, in_height, in_width,
                                                       resize),
                  pipe="fp32",
                  nbytes=_src_bytes(srcs, n) + 4.0 * n * (oh * ow * out_channels + sum(
                      s.aux_channels * in_height * in_width for s in srcs if s.kind == _lib.KBN_SRC_XYZ))), "kbn_conv2d_forward")
    return out


# ------------------------------------------------- layer-by-layer form: activation in place, backprojection
ACT_KINDS = {"elu": _lib.KBN_ACT_ELU, "sigmoid": _lib.KBN_ACT_SIGMOID}


@_on_tensor_device
def activation_(t: torch.Tensor, kind: str, out_absmax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """t <- ELU(t) / sigmoid(t) in place (kbn_activation_forward): the activation of a conv that was launched without one, for
    the activations the fused kernels do not carry (reference src/net_utils.py:38-43).  `t`: a dense N x C x H x W tensor or a
    channel slice of one.  `out_absmax`: the tensor's per-frame max |a| slot (ActStats), filled with the maxima of the activated values."""
    lib = _lib.load()
    tptr, tbs = _planes(t, "t")
    n = t.shape[0]
    per = t.shape[1] * t.shape[2] * t.shape[3]
    check(_launch("activation", 0.0, lambda: lib.kbn_activation_forward(tptr, tbs, n, per, ACT_KINDS[kind], _slot_ptr(out_absmax, n), _stream()),
                  nbytes=8.0 * n * per), "kbn_activation_forward")
    return t


@_on_tensor_device
def scale_planes(x: torch.Tensor, z: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[n, c] = x[n, c] * z[n, 0] (kbn_scale_planes_forward): the KB block's xyz = coordinates * z (reference
    src/net_utils.py:1357-1359) when z is a tensor of its own."""
    lib = _lib.load()
    xptr, xbs = _planes(x, "x")
    zptr, zbs = _planes(z, "z")
    n, c, h, w = x.shape
    if tuple(z.shape) != (n, 1, h, w):
        raise KbnError(f"scale_planes: z has shape {tuple(z.shape)}, expected {(n, 1, h, w)}")
    if out is None:
        out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    optr, obs = _planes(out, "out")
    if tuple(out.shape) != (n, c, h, w):
        raise KbnError(f"scale_planes: out has shape {tuple(out.shape)}, expected {(n, c, h, w)}")
    check(_launch("scale_planes", 1.0 * n * c * h * w,
                  lambda: lib.kbn_scale_planes_forward(xptr, xbs, zptr, zbs, optr, obs, n, c, h, w, _stream()),
                  nbytes=4.0 * n * h * w * (2 * c + 1)), "kbn_scale_planes_forward")
    return out


# ---------------------------------------------------------------------- up-conv 2x
def _deconv_weight(weight: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d's in x out x 3 x 3 parameter with out_channels leading: the layout the packers read."""
    return weight.detach().permute(1, 0, 2, 3).contiguous()


@_on_tensor_device
def pack_upconv2x_weight(weight: torch.Tensor, out: Optional[torch.Tensor] = None, transposed: bool = False) -> torch.Tensor:
    """OIHW 3x3 weight -> phase-summed MFMA blob for `upconv2x` (once per weight).  `transposed`: `weight` is a
    ConvTranspose2d(kernel 3, stride 2, padding 1, output_padding 1) parameter, in x out x 3 x 3, and the blob is
    for `upconv2x(transposed=True)` (kbn_deconv2x_pack_weight)."""
    lib = _lib.load()
    w = _deconv_weight(weight) if transposed else weight.detach().contiguous()
    _require(w, "weight", 4)
    oc, cin, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise KbnError("upconv2x needs a 3x3 weight")
    nfloats = lib.kbn_upconv2x_packed_weight_bytes(oc, cin) // 4
    packed = out if _reusable(out, nfloats, w) else torch.empty(nfloats, device=w.device, dtype=torch.float32)
    if transposed:
        check(lib.kbn_deconv2x_pack_weight(w.data_ptr(), packed.data_ptr(), oc, cin, _stream()), "kbn_deconv2x_pack_weight")
    else:
        check(lib.kbn_upconv2x_pack_weight(w.data_ptr(), packed.data_ptr(), oc, cin, _stream()),
              "kbn_upconv2x_pack_weight")
    return packed


@_on_tensor_device
def upconv2x(x: torch.Tensor, packed_weight: torch.Tensor, out_channels: int, out: torch.Tensor,
             negative_slope: Optional[float] = 0.2, out_absmax: Optional[torch.Tensor] = None, transposed: bool = False):
    """nearest 2x upsample + conv3x3 (+ LeakyReLU): x N x C x h x w -> out N x out_channels x 2h x 2w.
    `transposed`: ConvTranspose2d(kernel 3, stride 2, padding 1, output_padding 1) (+ LeakyReLU) instead, same sizes
    (kbn_deconv2x_forward; blob from pack_upconv2x_weight(transposed=True))."""
    lib = _lib.load()
    xptr, xbs = _planes(x, "x")
    n, cin, h, w = x.shape
    optr, obs = _planes(out, "out")
    if tuple(out.shape) != (n, out_channels, 2 * h, 2 * w):
        raise KbnError(f"out has shape {tuple(out.shape)}, expected {(n, out_channels, 2 * h, 2 * w)}")
    fwd = lib.kbn_deconv2x_forward if transposed else lib.kbn_upconv2x_forward
    name = "kbn_deconv2x_forward" if transposed else "kbn_upconv2x_forward"
    # algorithmic work of the reference formulation (9 taps at the upsampled resolution; transposed: 9 taps per SOURCE pixel)
    work = 2.0 * n * h * w * cin * 9 * out_channels * (1 if transposed else 4)
    if transposed:   # always the four-phase form: 16 channel products per source pixel over the padded channel counts
        executed = lambda: upconv2x_executed_flops(n, cin, out_channels, h, w, plain=True)
    else:
        executed = lambda: upconv2x_executed_flops(n, cin, out_channels, h, w)
    check(_launch("conv_up2x", work,
                  lambda: fwd(xptr, xbs, packed_weight.data_ptr(), optr, obs, n, cin,
                              out_channels, h, w, 0 if negative_slope is None else 1,
                              0.0 if negative_slope is None else float(negative_slope),
                              _slot_ptr(out_absmax, n), _stream()),
                  executed=executed, pipe="fp32",
                  nbytes=4.0 * n * h * w * (cin + 4 * out_channels)), name)
    return out


# ------------------------------------------------------------------------ KB block
@_on_tensor_device
def kb_block(image, depth, coordinates, kinv, fused, packed_w_image, packed_w_depth, proj_weight,
             packed_w_fused, filters_image: int, filters_depth: int, filters_fused: int,
             out_image, out_depth, out_fused, negative_slope: float = 0.2, absmax_image=None, absmax_depth=None,
             absmax_fused=None):
    """Inputs/outputs may be channel slices; exactly one of coordinates / kinv may be None.  absmax_*: slots that
    receive max |out| per frame of the three outputs (two may be the same slot)."""
    lib = _lib.load()
    n, ci, h, w = image.shape
    iptr, ibs = _planes(image, "image")
    dptr, dbs = _planes(depth, "depth")
    cd = depth.shape[1]
    if fused is not None:
        fptr, fbs = _planes(fused, "fused")
        cf = fused.shape[1]
    else:
        fptr, fbs, cf = None, 0, 0
    cptr = None
    if coordinates is not None:
        _require(coordinates, "coordinates", 4)
        if not coordinates.is_contiguous() or tuple(coordinates.shape) != (n, 3, h, w):
            raise KbnError("coordinates must be a contiguous N x 3 x H x W tensor")
        cptr = coordinates.data_ptr()
    kptr = None
    if kinv is not None:
        _require(kinv, "kinv", 3)
        kptr = kinv.data_ptr()
    pw = proj_weight.detach().contiguous()
    _require(pw, "proj_weight")
    oh, ow = (h + 1) // 2, (w + 1) // 2
    for t, f, nm in ((out_image, filters_image, "out_image"), (out_depth, filters_depth, "out_depth"),
                     (out_fused, filters_fused, "out_fused")):
        if tuple(t.shape) != (n, f, oh, ow):
            raise KbnError(f"{nm} has shape {tuple(t.shape)}, expected {(n, f, oh, ow)}")
    oi, oibs = _planes(out_image, "out_image")
    od, odbs = _planes(out_depth, "out_depth")
    of, ofbs = _planes(out_fused, "out_fused")
    flops = 2.0 * n * oh * ow * (9 * ci * filters_image + 9 * (cd + 3) * filters_depth +
                                  (ci + 3 + cf) * filters_fused)
    check(_launch("kb_block", flops,
                  lambda: lib.kbn_kb_block_forward(iptr, ibs, dptr, dbs, cptr, kptr, fptr, fbs,
                                                   packed_w_image.data_ptr(), packed_w_depth.data_ptr(),
                                                   pw.data_ptr(), packed_w_fused.data_ptr(), oi, oibs, od, odbs,
                                                   of, ofbs, n, h, w, ci, cd, cf, filters_image, filters_depth,
                                                   filters_fused, float(negative_slope), _slot_ptr(absmax_image, n),
                                                   _slot_ptr(absmax_depth, n), _slot_ptr(absmax_fused, n), _stream()),
                  executed=flops,   # direct convs: executed = algorithmic (tile padding not counted)
                  pipe="fp32", nbytes=4.0 * n * (h * w * (ci + cd + cf) + oh * ow * (filters_image + filters_depth + filters_fused))),
          "kbn_kb_block_forward")
    return out_image, out_depth, out_fused


# ---------------------------------------------------------------------- depth head
HEAD_MAX_CHANNELS = 16   # csrc/head.hip HD_MAXC


@_on_tensor_device
def depth_head(x, weight, min_predict_depth: float, max_predict_depth: float, return_logits=False, out=None):
    """`out`: optional contiguous N x 1 x H x W destination (e.g. a batch slice of a larger output buffer)."""
    lib = _lib.load()
    _require(x, "x", 4)
    x = x.contiguous()
    w = weight.detach().contiguous()
    _require(w, "weight", 4)
    n, c, h, wd = x.shape
    if tuple(w.shape) != (1, c, 3, 3):
        raise KbnError(f"depth head weight must be 1 x {c} x 3 x 3")
    if c > HEAD_MAX_CHANNELS:
        # run_kbnet.py --n_filters_decoder with a last width past the head kernel's 16 channels: output0 as a conv of its own, then the
        # mapping as the head kernel over that one plane with an identity tap (1.0 * logit + eight exact zeros: the logits as they are)
        plane = conv2d([tensor_src(x, "x")], pack_conv_weight(w, 1), n, 1, 3, 1, h, wd,
                       torch.empty((n, 1, h, wd), device=x.device, dtype=torch.float32), negative_slope=None)
        # (device ops only -- a fill and a pad: this runs under HIP-graph capture too)
        ident = torch.nn.functional.pad(torch.ones((1, 1, 1, 1), device=x.device, dtype=torch.float32), (1, 1, 1, 1))
        return depth_head(plane, ident, min_predict_depth, max_predict_depth, return_logits=return_logits, out=out)
    if out is None:
        depth = torch.empty((n, 1, h, wd), device=x.device, dtype=torch.float32)
    else:
        _require(out, "out", 4)
        if tuple(out.shape) != (n, 1, h, wd) or not out.is_contiguous():
            raise KbnError(f"out must be a contiguous {(n, 1, h, wd)} tensor")
        depth = out
    logits = torch.empty_like(depth) if return_logits else None
    check(_launch("depth_head", 4.0 * n * h * wd * (c + 1),
                  lambda: lib.kbn_depth_head_forward(x.data_ptr(), w.data_ptr(), depth.data_ptr(),
                                                     logits.data_ptr() if return_logits else None, n, c, h, wd,
                                                     float(min_predict_depth), float(max_predict_depth),
                                                     _stream()), nbytes=4.0 * n * h * wd * (c + 1)), "kbn_depth_head_forward")
    return (depth, logits) if return_logits else depth


@_on_tensor_device
def conv_head(x, w_conv, w_out, min_predict_depth: float, max_predict_depth: float,
              negative_slope: Optional[float] = 0.2, return_logits=False, out=None):
    """deconv0's second conv + output0 + depth mapping in one launch (kbn_conv_head_forward).  Returns None when
    the shape does not qualify (channels % 4, width % 4, alignment): the caller runs the two-launch path."""
    lib = _lib.load()
    xptr, xbs = _planes(x, "x")
    n, c, h, wd = x.shape
    wc = w_conv.detach().contiguous()
    wo = w_out.detach().contiguous()
    _require(wc, "w_conv", 4)
    _require(wo, "w_out", 4)
    if tuple(wc.shape) != (c, c, 3, 3) or tuple(wo.shape) != (1, c, 3, 3):
        raise KbnError(f"conv_head weights must be {c} x {c} x 3 x 3 and 1 x {c} x 3 x 3")
    if out is None:
        depth = torch.empty((n, 1, h, wd), device=x.device, dtype=torch.float32)
    else:
        _require(out, "out", 4)
        if tuple(out.shape) != (n, 1, h, wd) or not out.is_contiguous():
            raise KbnError(f"out must be a contiguous {(n, 1, h, wd)} tensor")
        depth = out
    logits = torch.empty_like(depth) if return_logits else None
    flops = 2.0 * n * h * wd * c * 9 * c
    status = _launch("conv_head", flops,
                     lambda: lib.kbn_conv_head_forward(xptr, xbs, wc.data_ptr(), wo.data_ptr(), depth.data_ptr(),
                                                       logits.data_ptr() if return_logits else None, n, c, h, wd,
                                                       0 if negative_slope is None else 1,
                                                       0.0 if negative_slope is None else float(negative_slope),
                                                       float(min_predict_depth), float(max_predict_depth), _stream()),
                     # 75 m-blocks of 16 positions per 64 x 16 tile, 16 filter columns, K = 9 c
                     executed=2.0 * n * (-(-h // 16)) * (-(-wd // 64)) * 75 * 16 * 16 * 9 * c, pipe="fp32",
                     nbytes=4.0 * n * h * wd * (c + 1))
    if status == _lib.KBN_ERR_UNSUPPORTED:
        if PROFILE is not None:
            PROFILE.pop()
        return None
    check(status, "kbn_conv_head_forward")
    return (depth, logits) if return_logits else depth


@_on_tensor_device
def pack_conv_tail_weight(w_conv: torch.Tensor, out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Blob of `conv_tail` from the raw C x C x 3 x 3 weight of deconv0's second conv (kbn_conv_tail_pack_weight); None when
    C is outside the kernel's range."""
    lib = _lib.load()
    w = w_conv.detach().contiguous()
    _require(w, "w_conv", 4)
    c = w.shape[0]
    if tuple(w.shape) != (c, c, 3, 3):
        return None
    nbytes = lib.kbn_conv_tail_packed_weight_bytes(c)
    if nbytes == 0:
        return None
    packed = out if _reusable(out, nbytes // 4, w) else torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    check(lib.kbn_conv_tail_pack_weight(w.data_ptr(), packed.data_ptr(), c, _stream()), "kbn_conv_tail_pack_weight")
    return packed


@_on_tensor_device
def conv_tail(x, packed_w_conv, w_out, min_predict_depth: float, max_predict_depth: float,
              negative_slope: Optional[float] = 0.2, return_logits=False, out=None):
    """deconv0's second conv (on split fp16 operands) + output0 + depth mapping in one launch (kbn_conv_tail_forward).
    `x`: N x C x H x W fp32, or the up-conv's PairTensor of 16 channels (kbn_conv_tail_forward_pair; C = w_out's channels).
    Returns None when the shape does not qualify: the caller runs conv_head / the two-launch path."""
    lib = _lib.load()
    wo = w_out.detach().contiguous()
    _require(wo, "w_out", 4)
    pair = isinstance(x, PairTensor)
    if pair:
        n, cp, h, wd = x.shape
        c = wo.shape[1]
        if cp != 16 or c > 16:
            return None
    else:
        xptr, xbs = _planes(x, "x")
        n, c, h, wd = x.shape
    if tuple(wo.shape) != (1, c, 3, 3):
        raise KbnError(f"conv_tail: w_out must be 1 x {c} x 3 x 3")
    if out is None:
        depth = torch.empty((n, 1, h, wd), device=x.device, dtype=torch.float32)
    else:
        _require(out, "out", 4)
        if tuple(out.shape) != (n, 1, h, wd) or not out.is_contiguous():
            raise KbnError(f"out must be a contiguous {(n, 1, h, wd)} tensor")
        depth = out
    logits = torch.empty_like(depth) if return_logits else None
    flops = 2.0 * n * h * wd * c * 9 * c
    tiles = n * (-(-h // 16)) * (-(-wd // 32))
    tail_args = (packed_w_conv.data_ptr(), wo.data_ptr(), depth.data_ptr(), logits.data_ptr() if return_logits else None, n, c, h, wd,
                 0 if negative_slope is None else 1, 0.0 if negative_slope is None else float(negative_slope),
                 float(min_predict_depth), float(max_predict_depth))
    status = _launch("conv_tail", flops,
                     (lambda: lib.kbn_conv_tail_forward_pair(x.data.data_ptr(), x.data.stride(0), x.scale.data_ptr(), *tail_args, _stream()))
                     if pair else (lambda: lib.kbn_conv_tail_forward(xptr, xbs, *tail_args, _stream())),
                     executed=tiles * 39 * 15 * 2.0 * 16 * 16 * 32,   # 39 pixel blocks x 15 MFMAs of 16 x 16 x 32 per tile
                     pipe="fp16", nbytes=4.0 * n * h * wd * ((16 if pair else c) + 1))
    if status == _lib.KBN_ERR_UNSUPPORTED:
        if PROFILE is not None:
            PROFILE.pop()
        return None
    check(status, "kbn_conv_tail_forward")
    return (depth, logits) if return_logits else depth


# ----------------------------------------------------- fp32-grade convs on the 16-bit matrix core (split operands)
@_on_tensor_device
def pack_conv3x3_split_weight(weight: torch.Tensor, out: Optional[torch.Tensor] = None, stride: int = 1,
                              folded_up2x: bool = False, transposed: bool = False) -> torch.Tensor:
    """OIHW fp32 3x3 weight -> per-filter scaled two-term fp16 split in MFMA order for `conv3x3_split` with the same
    `stride` / `folded_up2x` (in_channels % 16 == 0).  `folded_up2x`: the sixteen 2x2 parity weights of the nearest-2x
    up-conv (sums of the 3x3 taps that read the same source pixel) instead of the nine taps.  `transposed`: `weight` is a
    ConvTranspose2d(kernel 3, stride 2, padding 1, output_padding 1) parameter (in x out x 3 x 3) and the blob is the one
    `conv3x3_split(up2x=True, folded_up2x=True, transposed=True)` reads (mode 4: the folded kernels, the layer's own taps)."""
    mode = 4 if transposed else (3 if folded_up2x else (2 if stride == 2 else 0))
    lib = _lib.load()
    w = _deconv_weight(weight) if transposed else weight.detach().contiguous()
    _require(w, "weight", 4)
    oc, cin, kh, kw = w.shape
    nbytes = lib.kbn_conv3x3_split_packed_weight_bytes(oc, cin, mode) if (kh, kw) == (3, 3) else 0
    if nbytes == 0:
        raise KbnError(f"conv3x3_split needs a 3x3 weight with in_channels % 16 == 0, got {tuple(w.shape)}")
    packed = out if _reusable(out, nbytes // 4, w) else torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    check(lib.kbn_conv3x3_split_pack_weight(w.data_ptr(), packed.data_ptr(), oc, cin, mode, _stream()), "kbn_conv3x3_split_pack_weight")
    return packed


def act_exponent_for(amax: float) -> int:
    """The exponent the split kernels derive on the device from a frame's max |a| (sp_act_scale, csrc/conv_split.hip):
    k = 14 - floor(log2(max |a|)) puts the maximum in [2^14, 2^15) of the fp16 window (overflow at 65504); zero or
    denormal maxima take 100, Inf / NaN -100.  Host-side restatement for tests and for callers of the static
    `act_exponent` argument."""
    import struct
    b = struct.unpack("<I", struct.pack("<f", abs(amax)))[0] if amax == amax else 0x7fc00000
    return max(-100, min(100, 14 + 127 - (b >> 23)))


def conv3x3_split_executed_flops(n, cin, out_channels, height, width, stride=1, folded_up2x=False):
    """fp16 MFMA FLOPs the split kernels ISSUE (what SQ_INSTS_MFMA x 32768 counts, profiles/*/traffic.json): three
    products per fp32 product over M = 32 output pixels of a row x N = 32 filters (16 for the narrow folded up-conv)
    x K = 16 channels per instruction.  Padding that is issued: the columns of the last 32-pixel segment of a row and
    the filters of the last 32-filter block.  Padding that is NOT issued since the kernels skip it (wave-uniform tests,
    csrc/conv_split.hip): output rows below the map and 32-filter blocks that lie entirely past the last filter of a
    64- / 128-wide tile.  Folded up-conv: 16 products per low-resolution pixel instead of 36 per output pixel quad."""
    if folded_up2x:
        nblk = 16 if (out_channels <= 16 and cin % 32 == 0) else 32   # narrow layers: v_mfma_f32_16x16x32_f16, 16-pixel segments
        seg = 16 if nblk == 16 else 32
        sh, sw = height // 2, width // 2
        wide = out_channels >= 64 and (-(-out_channels // 32)) % 2 == 0     # the 64-filter tiles (csrc/conv_split.hip, uf_wide)
        if nblk == 32 and wide and sw >= 32 and 1 <= sw % 32 <= 16:
            px = sh * 32 * (sw // 32) + -(-sh // 2) * 32                     # the narrow last column on transposed tiles: blocks of two rows
        else:
            px = sh * (-(-sw // seg) * seg)
        return 3 * 2.0 * n * px * cin * 16 * (-(-out_channels // nblk) * nblk)
    if stride == 1 and width >= 32 and 1 <= width % 32 <= 16:
        # the narrow last column runs on transposed tiles (32 rows x 16 columns; csrc/conv_split.hip, TP): blocks of two rows
        px = height * 32 * (width // 32) + -(-height // 2) * 32
    else:
        px = height * (-(-width // 32) * 32)
    return 3 * 2.0 * n * px * cin * 9 * (-(-out_channels // 32) * 32)


@_on_tensor_device
def conv3x3_split(srcs: List[ConvSrc], packed_weight: torch.Tensor, n: int, out_channels: int, height: int, width: int,
                  out: torch.Tensor, up2x: bool = False, negative_slope: Optional[float] = 0.2, stride: int = 1,
                  act_exponent: int = -6, folded_up2x: bool = False, out_absmax: Optional[torch.Tensor] = None,
                  transposed: bool = False, ksplit: int = 1):
    """3x3 conv (+ LeakyReLU) of up to two concatenated tensor sources (`up2x`: of ONE source upsampled 2x, nearest;
    `stride` 2: sources are the 2x larger input planes), fp32 in / fp32 out, every product taken as three fp16 MFMAs
    over two-term splits of both operands (kbn_conv3x3_split_forward): fp32-grade accuracy at 3/16 of the fp32 MFMA's
    time.  `height` x `width` is the OUTPUT size.  The fp16 window follows the data when every source carries its absmax
    slot (tensor_src(absmax=)): per frame, on the device; only sources without slots fall back to the static
    `act_exponent` k (|a| 2^k < 65504).  `out_absmax`: slot that receives max |out| per frame.
    `folded_up2x` (with `up2x`): the folded 16-product form, weights from
    pack_conv3x3_split_weight(folded_up2x=True).  `transposed` (with both): the layer is a ConvTranspose2d(kernel 3, stride 2,
    padding 1, output_padding 1) instead -- same kernels, blob from pack_conv3x3_split_weight(transposed=True).
    `ksplit` > 1: the latency form of the plain, the stride-2 and the wide folded up-conv (kbn_conv3x3_split_forward_ksplit): every
    tile's K loop spread over `ksplit` workgroups, their partial sums added in split order by a second kernel -- for launches too small
    to fill the chip (ksplit_for); fp32 sources and an fp32 output only.  Returns None when the shape does not qualify (the caller
    stays on the fp32-MFMA kernels)."""
    if transposed and not (up2x and folded_up2x):
        raise KbnError("conv3x3_split: transposed goes with up2x and folded_up2x")
    if up2x and stride != 1:
        raise KbnError("conv3x3_split: up2x and stride 2 are mutually exclusive")
    lib = _lib.load()
    arr = (ConvSrc * len(srcs))(*srcs)
    pair = isinstance(out, PairTensor)   # the output in the producer-written split format (its slot doubles as out_absmax)
    if pair:
        optr, obs = (None, 0) if (out.sub is None or stride != 2) else _planes(out.sub, "out.sub")
        if out_absmax is None:
            out_absmax = out.absmax
    else:
        optr, obs = _planes(out, "out")
    want = (n, 16 if (pair and up2x and folded_up2x and out_channels <= 16) else out_channels, height, width)   # the narrow up-conv writes 16
    if tuple(out.shape) != want:
        raise KbnError(f"out has shape {tuple(out.shape)}, expected {want}")
    cin = sum(s.channels for s in srcs)
    flops = 2.0 * n * height * width * cin * 9 * out_channels / (4 if transposed else 1)   # transposed: nine taps per SOURCE pixel
    mode = (4 if transposed else (3 if folded_up2x else 1)) if up2x else (2 if stride == 2 else 0)
    if ksplit > 1:
        if pair or mode == 1:
            raise KbnError("conv3x3_split: ksplit goes with the plain / stride-2 / folded up-conv forms and an fp32 output")
        ws = torch.empty((ksplit, n, out_channels, height, width), device=out.device, dtype=torch.float32)
        status = _launch(("conv_split", "", "conv_split_s2", "conv_split_upfold", "conv_split_upfold")[mode], flops,
                         lambda: lib.kbn_conv3x3_split_forward_ksplit(arr, len(srcs), packed_weight.data_ptr(), optr, obs, n, out_channels, height, width,
                                                                      mode, max(-60, min(60, int(act_exponent))), 0 if negative_slope is None else 1,
                                                                      0.0 if negative_slope is None else float(negative_slope),
                                                                      _slot_ptr(out_absmax, n), int(ksplit), ws.data_ptr(), _stream()),
                         executed=conv3x3_split_executed_flops(n, cin, out_channels, height, width, stride, up2x and folded_up2x),
                         pipe="fp16", nbytes=_src_bytes(srcs, n) + 4.0 * n * height * width * out_channels * (1 + 2 * ksplit))
        if status == _lib.KBN_ERR_UNSUPPORTED:
            if PROFILE is not None:
                PROFILE.pop()
            return None
        check(status, "kbn_conv3x3_split_forward_ksplit")
        return out
    status = _launch(("conv_split", "conv_split_up", "conv_split_s2", "conv_split_upfold", "conv_split_upfold")[mode], flops,
                     lambda: lib.kbn_conv3x3_split_forward(arr, len(srcs), packed_weight.data_ptr(), optr, obs, n,
                                                           out_channels, height, width,
                                                           mode, max(-60, min(60, int(act_exponent))),
                                                           0 if negative_slope is None else 1,
                                                           0.0 if negative_slope is None else float(negative_slope),
                                                           _slot_ptr(out_absmax, n),
                                                           out.data.data_ptr() if pair else None,
                                                           out.data.stride(0) if pair else 0,
                                                           out.scale.data_ptr() if pair else None, _stream()),
                     executed=conv3x3_split_executed_flops(n, cin, out_channels, height, width, stride, up2x and folded_up2x),
                     pipe="fp16", nbytes=_src_bytes(srcs, n) + 4.0 * n * height * width * want[1]
                     + (4.0 * n * out_channels * out.sub.shape[2] * out.sub.shape[3] if (pair and out.sub is not None and stride == 2) else 0.0))
    if status == _lib.KBN_ERR_UNSUPPORTED:
        if PROFILE is not None:
            PROFILE.pop()
        return None
    check(status, "kbn_conv3x3_split_forward")
    return out


def ksplit_for(cin: int, out_channels: int, height: int, width: int, stride: int = 1, cus: int = 256, up2x: bool = False, frames: int = 1) -> int:
    """How many workgroups share a tile's K loop in the latency form (KBNetModel.set_latency_mode(frames=)): as many as bring a launch
    of `frames` frames to about one workgroup per CU -- every range at least two 16-channel chunks long, at most 16 -- and 1 when
    its tiles already fill half the chip or the output is large (the partial planes cost HBM traffic: 2 x ksplit x the output).  A
    function of the layer and of the MODE's `frames`, never of the batch a call happens to carry: inside a mode a frame's bits do not
    depend on what runs beside it."""
    if up2x:   # the folded up-conv: 8 x 32 low-resolution pixels x 64 filters per workgroup
        tiles = -(-(width // 2) // 32) * (-(-(height // 2) // 8)) * (-(-out_channels // 64))
    else:
        tiles = -(-width // 32) * (-(-height // (8 if stride == 2 else 16))) * (-(-out_channels // (128 if stride == 2 else 64)))
    tiles *= max(1, int(frames))
    chunks = cin // 16
    if tiles * 2 > cus or chunks < 4 or 4 * out_channels * height * width * max(1, int(frames)) > (16 << 20):
        return 1
    ks = min(16, chunks // 2, max(1, cus // tiles))
    while ks > 1 and -(-chunks // ks) * (ks - 1) >= chunks:   # no empty range
        ks -= 1
    return ks


# ----------------------------------------------------- KB block: conv_fused on split operands
@_on_tensor_device
def pack_conv1x1s2_split_weight(weight: torch.Tensor, xyz_offset: int = -1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """OI11 fp32 weight of a 1x1 conv -> blob of `conv1x1s2_split`: per-filter scaled two-term fp16 split of the tensor
    channels in MFMA order (+ the fp32 weights of the three xyz channels that start at input channel `xyz_offset`,
    -1: the conv has none).  Tensor channels (in_channels, minus 3 with xyz) % 16 == 0."""
    lib = _lib.load()
    w = weight.detach().contiguous()
    _require(w, "weight", 4)
    oc, cin, kh, kw = w.shape
    has_xyz = xyz_offset >= 0
    nbytes = lib.kbn_conv1x1s2_split_packed_weight_bytes(oc, cin - (3 if has_xyz else 0), 1 if has_xyz else 0) if (kh, kw) == (1, 1) else 0
    if nbytes == 0:
        raise KbnError(f"conv1x1s2_split needs a 1x1 weight with tensor channels % 16 == 0, got {tuple(w.shape)}")
    packed = out if _reusable(out, nbytes // 4, w) else torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    check(lib.kbn_conv1x1s2_split_pack_weight(w.data_ptr(), packed.data_ptr(), oc, cin, int(xyz_offset), _stream()),
          "kbn_conv1x1s2_split_pack_weight")
    return packed


@_on_tensor_device
def kb_xyz_s2(depth: torch.Tensor, proj_weight: torch.Tensor, kinv: torch.Tensor, negative_slope: Optional[float] = 0.2,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The KB block's backprojection K^-1 [x y 1]^T z, z = act(proj_weight . depth), at the input pixels (2y, 2x) its
    stride-2 1x1 conv reads: N x 3 x ceil(H/2) x ceil(W/2) (kbn_kb_xyz_s2_forward)."""
    lib = _lib.load()
    dptr, dbs = _planes(depth, "depth")
    n, cd, h, w = depth.shape
    oh, ow = (h + 1) // 2, (w + 1) // 2
    pw = proj_weight.detach().contiguous()
    if pw.numel() != cd or tuple(kinv.shape) != (n, 3, 3) or not kinv.is_contiguous():
        raise KbnError("kb_xyz_s2: proj_weight must hold one weight per depth channel, kinv must be a dense N x 3 x 3")
    if out is None:
        out = torch.empty((n, 3, oh, ow), device=depth.device, dtype=torch.float32)
    optr, obs = _planes(out, "xyz")
    check(_launch("kb_xyz", 2.0 * n * oh * ow * cd,
                  lambda: lib.kbn_kb_xyz_s2_forward(dptr, dbs, cd, h, w, pw.data_ptr(), kinv.data_ptr(),
                                                    0 if negative_slope is None else 1,
                                                    0.0 if negative_slope is None else float(negative_slope),
                                                    optr, obs, n, _stream()), nbytes=4.0 * n * oh * ow * (cd + 3)), "kbn_kb_xyz_s2_forward")
    return out


@_on_tensor_device
def conv1x1s2_split(srcs: List[ConvSrc], packed_weight: torch.Tensor, xyz: Optional[torch.Tensor], n: int, out_channels: int,
                    height: int, width: int, out: torch.Tensor, negative_slope: Optional[float] = 0.2, act_exponent: int = -6,
                    out_absmax: Optional[torch.Tensor] = None):
    """1x1 stride-2 conv (+ LeakyReLU) of one or two tensor sources (+ the three fp32 xyz channels of `xyz`, from
    kb_xyz_s2) on split operands (kbn_conv1x1s2_split_forward); `height` x `width` is the OUTPUT size.  None when the
    shape does not qualify."""
    lib = _lib.load()
    arr = (ConvSrc * len(srcs))(*srcs)
    optr, obs = _planes(out, "out")
    if tuple(out.shape) != (n, out_channels, height, width):
        raise KbnError(f"out has shape {tuple(out.shape)}, expected {(n, out_channels, height, width)}")
    xptr, xbs = (None, 0)
    if xyz is not None:
        if tuple(xyz.shape) != (n, 3, height, width):
            raise KbnError(f"xyz has shape {tuple(xyz.shape)}, expected {(n, 3, height, width)}")
        xptr, xbs = _planes(xyz, "xyz")
    cin = sum(s.channels for s in srcs)
    flops = 2.0 * n * height * width * (cin + (3 if xyz is not None else 0)) * out_channels
    executed = 3 * 2.0 * n * (-(-height // 8) * 8) * (-(-width // 32) * 32) * cin * (-(-out_channels // 128) * 128)   # this kernel skips nothing
    status = _launch("conv_split_1x1s2", flops,
                     lambda: lib.kbn_conv1x1s2_split_forward(arr, len(srcs), packed_weight.data_ptr(), xptr, xbs, optr, obs, n,
                                                             out_channels, height, width, max(-60, min(60, int(act_exponent))),
                                                             0 if negative_slope is None else 1,
                                                             0.0 if negative_slope is None else float(negative_slope),
                                                             _slot_ptr(out_absmax, n), _stream()), executed=executed, pipe="fp16",
                     # a 1x1 stride-2 conv needs the even pixels of its sources only
                     nbytes=4.0 * n * height * width * (cin + (3 if xyz is not None else 0) + out_channels))
    if status == _lib.KBN_ERR_UNSUPPORTED:
        if PROFILE is not None:
            PROFILE.pop()
        return None
    check(status, "kbn_conv1x1s2_split_forward")
    return out


# ----------------------------------------------------- encoder front: conv0_image + KB1's conv_image / conv_fused in one launch
@_on_tensor_device
def pack_kb1_front_weight(w_conv0: torch.Tensor, w_conv_image: torch.Tensor, w_conv_fused: torch.Tensor,
                          out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Blob of `kb1_front` (kbn_kb1_front_pack_weight) from conv0_image's weight (F0 x C x 3 x 3), the level-0 KB block's
    conv_image weight (FI x F0 x 3 x 3) and conv_fused weight (FI x (F0 + 3) x 1 x 1).  None when the shapes are outside
    the kernel's (F0 = FI = 48, C <= 4)."""
    lib = _lib.load()
    w0, wi, wf = (w.detach().contiguous() for w in (w_conv0, w_conv_image, w_conv_fused))
    for w, nm in ((w0, "w_conv0"), (wi, "w_conv_image"), (wf, "w_conv_fused")):
        _require(w, nm, 4)
    f0, c = w0.shape[0], w0.shape[1]
    fi = wi.shape[0]
    if tuple(w0.shape[2:]) != (3, 3) or tuple(wi.shape) != (fi, f0, 3, 3) or tuple(wf.shape) != (fi, f0 + 3, 1, 1):
        return None
    nbytes = lib.kbn_kb1_front_packed_weight_bytes(c, f0, fi)
    if nbytes == 0:
        return None
    packed = out if _reusable(out, nbytes // 4, w0) else torch.empty(nbytes // 4, device=w0.device, dtype=torch.float32)
    check(lib.kbn_kb1_front_pack_weight(w0.data_ptr(), wi.data_ptr(), wf.data_ptr(), packed.data_ptr(), c, f0, fi, _stream()),
          "kbn_kb1_front_pack_weight")
    return packed


def kb1_front_supported(image_channels: int, conv0_filters: int, kb_filters: int, height: int, width: int,
                        conv0_negative_slope: float, depth_branch: bool = False) -> bool:
    """Would kb1_front (depth_branch: kb1_depth_front) take this problem?  (kbn_kb1_front_query / kbn_kb1_depth_front_query)"""
    lib = _lib.load()
    q = lib.kbn_kb1_depth_front_query if depth_branch else lib.kbn_kb1_front_query
    return q(int(image_channels), int(conv0_filters), int(kb_filters), int(height), int(width), float(conv0_negative_slope)) == _lib.KBN_OK


@_on_tensor_device
def pack_kb1_front_next_weight(w_conv_fused: torch.Tensor, image_channels: int, out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Blob of kb1_front's `next` stage (kbn_kb1_front_next_pack_weight) from the NEXT KB block's conv_fused weight,
    F x (image_channels + 3 + fused_channels) x 1 x 1.  None when the widths are outside the kernel's (48 + 3 + 48 -> 96)."""
    lib = _lib.load()
    wf = w_conv_fused.detach().contiguous()
    _require(wf, "w_conv_fused", 4)
    fo, cin = wf.shape[0], wf.shape[1]
    cf = cin - 3 - image_channels
    if tuple(wf.shape[2:]) != (1, 1) or cf < 0:
        return None
    nbytes = lib.kbn_kb1_front_next_packed_weight_bytes(int(image_channels), int(cf), int(fo))
    if nbytes == 0:
        return None
    packed = out if _reusable(out, nbytes // 4, wf) else torch.empty(nbytes // 4, device=wf.device, dtype=torch.float32)
    check(lib.kbn_kb1_front_next_pack_weight(wf.data_ptr(), packed.data_ptr(), int(image_channels), int(cf), int(fo), _stream()),
          "kbn_kb1_front_next_pack_weight")
    return packed


def kb1_front_next_supported(image_channels: int, conv0_filters: int, kb_filters: int, next_filters: int, height: int, width: int,
                             conv0_negative_slope: float) -> bool:
    """Would kb1_front take the next level's conv_fused along?  (kbn_kb1_front_next_query; KBN_NO_FRONT_NEXT=1 says no)"""
    return _lib.load().kbn_kb1_front_next_query(int(image_channels), int(conv0_filters), int(kb_filters), int(next_filters), int(height),
                                                int(width), float(conv0_negative_slope)) == _lib.KBN_OK


@_on_tensor_device
def kb1_front(image: torch.Tensor, packed_weight: torch.Tensor, xyz: Optional[torch.Tensor],
              conv0_filters: int, kb_filters: int, out_image: torch.Tensor, out_fused: torch.Tensor,
              conv0_negative_slope: float = 0.2, kb_negative_slope: float = 0.2, out_image_absmax=None, out_fused_absmax=None,
              next_fused=None):
    """conv0_image -> (conv_image, conv_fused) of the level-0 KB block in one launch, conv0's output kept on the CU
    (kbn_kb1_front_forward).  `xyz`: the backprojection channels from kb_xyz_s2.  None when the shape does not qualify.
    `next_fused` = (packed_next, xyz_next, out_next_fused, negative_slope, out_next_absmax): the NEXT KB level's conv_fused in the
    same launch (kbn_kb1_front_next_forward) -- its inputs are the even pixels of this launch's two outputs."""
    lib = _lib.load()
    iptr, ibs = _planes(image, "image")
    n, c, h, w = image.shape
    oh, ow = (h + 1) // 2, (w + 1) // 2
    for t, nm in ((out_image, "out_image"), (out_fused, "out_fused")):
        if tuple(t.shape) != (n, kb_filters, oh, ow):
            raise KbnError(f"{nm} has shape {tuple(t.shape)}, expected {(n, kb_filters, oh, ow)}")
    oi, oibs = _planes(out_image, "out_image")
    of, ofbs = _planes(out_fused, "out_fused")
    xptr, xbs = (None, 0)
    if xyz is not None:
        if tuple(xyz.shape) != (n, 3, oh, ow):
            raise KbnError(f"xyz has shape {tuple(xyz.shape)}, expected {(n, 3, oh, ow)}")
        xptr, xbs = _planes(xyz, "xyz")
    flops = 2.0 * n * (h * w * c * 9 * conv0_filters + oh * ow * conv0_filters * 9 * kb_filters + oh * ow * (conv0_filters + 3) * kb_filters)
    # issued fp16 MFMA FLOPs: per 8 x 16 tile and 16-filter chunk 36 x 9 (conv0) + 8 x 6 x 3 x 3 (conv_image, conv_fused) MFMAs of 16 x 16 x 32
    tiles = n * (-(-oh // 8)) * (-(-ow // 16))
    executed = tiles * (conv0_filters // 16) * (36 * 9 + 8 * 6 * (kb_filters // 16) * 3) * 2.0 * 16 * 16 * 32
    nbytes = 4.0 * n * (h * w * c + oh * ow * (2 * kb_filters + (3 if xyz is not None else 0)))
    if next_fused is None:
        call = lambda: lib.kbn_kb1_front_forward(iptr, ibs, packed_weight.data_ptr(), xptr, xbs,
                                                 oi, oibs, of, ofbs, n, c, conv0_filters, kb_filters, h, w,
                                                 float(conv0_negative_slope), float(kb_negative_slope),
                                                 _slot_ptr(out_image_absmax, n), _slot_ptr(out_fused_absmax, n), _stream())
    else:
        packed_next, xyz_next, out_next, slope_next, amax_next = next_fused
        h2, w2 = (oh + 1) // 2, (ow + 1) // 2
        fo = out_next.shape[1]
        if tuple(out_next.shape) != (n, fo, h2, w2) or tuple(xyz_next.shape) != (n, 3, h2, w2):
            raise KbnError(f"next_fused: out {tuple(out_next.shape)} / xyz {tuple(xyz_next.shape)}, expected {(n, fo, h2, w2)} / {(n, 3, h2, w2)}")
        on, onbs = _planes(out_next, "out_next_fused")
        xn, xnbs = _planes(xyz_next, "xyz_next")
        flops += 2.0 * n * h2 * w2 * (2 * kb_filters + 3) * fo
        executed += tiles * 2 * (fo // 16) * (2 * kb_filters // 32) * 3 * 2.0 * 16 * 16 * 32   # 2 pixel blocks x filter blocks x k-steps x 3 products
        nbytes += 4.0 * n * h2 * w2 * (fo + 3)
        call = lambda: lib.kbn_kb1_front_next_forward(iptr, ibs, packed_weight.data_ptr(), xptr, xbs,
                                                      oi, oibs, of, ofbs, n, c, conv0_filters, kb_filters, h, w,
                                                      float(conv0_negative_slope), float(kb_negative_slope),
                                                      _slot_ptr(out_image_absmax, n), _slot_ptr(out_fused_absmax, n),
                                                      packed_next.data_ptr(), xn, xnbs, on, onbs, fo, float(slope_next),
                                                      _slot_ptr(amax_next, n), _stream())
    status = _launch("kb1_front", flops, call, executed=executed, pipe="fp16", nbytes=nbytes)
    if status == _lib.KBN_ERR_UNSUPPORTED:
        if PROFILE is not None:
            PROFILE.pop()
        return None
    check(status, "kbn_kb1_front_forward")
    return out_image, out_fused


@_on_tensor_device
def pack_kb1_depth_front_weight(w_conv0: torch.Tensor, w_conv_depth: torch.Tensor, w_proj: torch.Tensor,
                                out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Blob of `kb1_depth_front` from conv0_depth's weight (16 x C x 3 x 3), the level-0 KB block's conv_depth weight
    (16 x 19 x 3 x 3) and proj_depth weight (1 x 16 x 1 x 1).  None when the shapes are outside the kernel's."""
    lib = _lib.load()
    w0, wc, wp = (w.detach().contiguous() for w in (w_conv0, w_conv_depth, w_proj))
    for w, nm in ((w0, "w_conv0"), (wc, "w_conv_depth"), (wp, "w_proj")):
        _require(w, nm, 4)
    f0, c = w0.shape[0], w0.shape[1]
    fd = wc.shape[0]
    if tuple(w0.shape[2:]) != (3, 3) or tuple(wc.shape) != (fd, f0 + 3, 3, 3) or wp.numel() != f0:
        return None
    nbytes = lib.kbn_kb1_depth_front_packed_weight_bytes(c, f0, fd)
    if nbytes == 0:
        return None
    packed = out if _reusable(out, nbytes // 4, w0) else torch.empty(nbytes // 4, device=w0.device, dtype=torch.float32)
    check(lib.kbn_kb1_depth_front_pack_weight(w0.data_ptr(), wc.data_ptr(), wp.data_ptr(), packed.data_ptr(), c, f0, fd, _stream()),
          "kbn_kb1_depth_front_pack_weight")
    return packed


@_on_tensor_device
def kb1_depth_front(depth: torch.Tensor, kinv: torch.Tensor, packed_weight: torch.Tensor, conv0_filters: int, kb_filters: int,
                    out_depth: torch.Tensor, conv0_negative_slope: float = 0.2, kb_negative_slope: float = 0.2,
                    proj_negative_slope: Optional[float] = 0.2, out_depth_absmax=None, xyz: Optional[torch.Tensor] = None):
    """conv0_depth -> conv_depth of the level-0 KB block (coordinate channels in fp32) and the backprojection channels xyz, in
    one launch (kbn_kb1_depth_front_forward).  Returns (out_depth, xyz) or None when the shape does not qualify."""
    lib = _lib.load()
    dptr, dbs = _planes(depth, "depth")
    n, c, h, w = depth.shape
    oh, ow = (h + 1) // 2, (w + 1) // 2
    if tuple(out_depth.shape) != (n, kb_filters, oh, ow):
        raise KbnError(f"out_depth has shape {tuple(out_depth.shape)}, expected {(n, kb_filters, oh, ow)}")
    if tuple(kinv.shape) != (n, 3, 3) or not kinv.is_contiguous():
        raise KbnError("kinv must be a dense N x 3 x 3")
    _require(kinv, "kinv", 3)
    if xyz is None:
        xyz = torch.empty((n, 3, oh, ow), device=depth.device, dtype=torch.float32)
    optr, obs = _planes(out_depth, "out_depth")
    xptr, xbs = _planes(xyz, "xyz")
    flops = 2.0 * n * (h * w * c * 9 * conv0_filters + oh * ow * ((conv0_filters + 3) * 9 * kb_filters + conv0_filters))
    tiles = n * (-(-oh // 8)) * (-(-ow // 16))
    executed = tiles * (36 * 9 + 8 * 15) * 2.0 * 16 * 16 * 32
    status = _launch("kb1_depth_front", flops,
                     lambda: lib.kbn_kb1_depth_front_forward(dptr, dbs, kinv.data_ptr(), packed_weight.data_ptr(), optr, obs, xptr, xbs,
                                                             n, c, conv0_filters, kb_filters, h, w, float(conv0_negative_slope),
                                                             float(kb_negative_slope), 0 if proj_negative_slope is None else 1,
                                                             0.0 if proj_negative_slope is None else float(proj_negative_slope),
                                                             _slot_ptr(out_depth_absmax, n), _stream()),
                     executed=executed, pipe="fp16", nbytes=4.0 * n * (h * w * c + oh * ow * (kb_filters + 3)))
    if status == _lib.KBN_ERR_UNSUPPORTED:
        if PROFILE is not None:
            PROFILE.pop()
        return None
    check(status, "kbn_kb1_depth_front_forward")
    return out_depth, xyz


# ----------------------------------------------------- S2D -> conv0_depth -> KB1's depth branch in one launch
@_on_tensor_device
def pack_s2d_depth_front_weight(w_pool_convs: List[torch.Tensor], w_conv: torch.Tensor, out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Blob of the on-chip S2D stage of `s2d_depth_front` (kbn_s2d_depth_front_pack_weight) from SparseToDensePool's weights:
    pool_convs.{0,1,2}.conv.weight and conv.conv.weight.  None when the shapes are outside the kernel's (three 1x1 layers of 8
    filters, a 3x3 conv over 8 + 2 channels)."""
    lib = _lib.load()
    if len(w_pool_convs) != 3:
        return None
    ws = [w.detach().contiguous() for w in w_pool_convs]
    wc = w_conv.detach().contiguous()
    for w in ws + [wc]:
        _require(w, "weight", 4)
    npool = ws[0].shape[1]
    if (tuple(ws[0].shape) != (8, npool, 1, 1) or tuple(ws[1].shape) != (8, 8, 1, 1) or tuple(ws[2].shape) != (8, 8, 1, 1)
            or tuple(wc.shape) != (8, 10, 3, 3)):
        return None
    nbytes = lib.kbn_s2d_depth_front_packed_weight_bytes(npool)
    if nbytes == 0:
        return None
    packed = out if _reusable(out, nbytes // 4, wc) else torch.empty(nbytes // 4, device=wc.device, dtype=torch.float32)
    check(lib.kbn_s2d_depth_front_pack_weight(ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), wc.data_ptr(), packed.data_ptr(), npool,
                                              _stream()), "kbn_s2d_depth_front_pack_weight")
    return packed


def s2d_depth_front_supported(input_channels: int, min_pool_sizes, max_pool_sizes, n_convolution: int, n_filter: int, conv0_filters: int,
                              kb_filters: int, height: int, width: int, s2d_negative_slope: float, conv0_negative_slope: float) -> bool:
    """Would `s2d_depth_front` take this problem?  (kbn_s2d_depth_front_query: pool preset, widths, slopes, switches)"""
    mins = [int(k) for k in min_pool_sizes if k > 1]
    maxs = [int(k) for k in max_pool_sizes if k > 1]
    return _lib.load().kbn_s2d_depth_front_query(int(input_channels), _int_array(mins), len(mins), _int_array(maxs), len(maxs), int(n_convolution),
                                                 int(n_filter), int(conv0_filters), int(kb_filters), int(height), int(width),
                                                 float(s2d_negative_slope), float(conv0_negative_slope)) == _lib.KBN_OK


@_on_tensor_device
def s2d_depth_front(x: torch.Tensor, kinv: torch.Tensor, packed_s2d: torch.Tensor, packed_weight: torch.Tensor, min_pool_sizes, max_pool_sizes,
                    conv0_filters: int, kb_filters: int, out_depth: torch.Tensor, s2d_negative_slope: float = 0.2,
                    conv0_negative_slope: float = 0.2, kb_negative_slope: float = 0.2, proj_negative_slope: Optional[float] = 0.2,
                    out_depth_absmax=None, xyz: Optional[torch.Tensor] = None):
    """SparseToDensePool -> conv0_depth -> conv_depth of the level-0 KB block and the backprojection channels xyz in ONE launch
    (kbn_s2d_depth_front_forward): `x` = N x 2 x H x W [sparse depth, validity]; the S2D tensor stays on the CU.  Returns
    (out_depth, xyz) or None when the problem does not qualify (the caller runs s2d_forward + kb1_depth_front)."""
    lib = _lib.load()
    xptr, xbs = _planes(x, "x")
    n, c, h, w = x.shape
    oh, ow = (h + 1) // 2, (w + 1) // 2
    if tuple(out_depth.shape) != (n, kb_filters, oh, ow):
        raise KbnError(f"out_depth has shape {tuple(out_depth.shape)}, expected {(n, kb_filters, oh, ow)}")
    if tuple(kinv.shape) != (n, 3, 3) or not kinv.is_contiguous():
        raise KbnError("kinv must be a dense N x 3 x 3")
    _require(kinv, "kinv", 3)
    if xyz is None:
        xyz = torch.empty((n, 3, oh, ow), device=x.device, dtype=torch.float32)
    optr, obs = _planes(out_depth, "out_depth")
    zptr, zbs = _planes(xyz, "xyz")
    mins = [int(k) for k in min_pool_sizes if k > 1]
    maxs = [int(k) for k in max_pool_sizes if k > 1]
    npool = len(mins) + len(maxs)
    nf = 8
    # the reference's work for the three layers (S2D at full resolution, conv0_depth, conv_depth + proj at half resolution)
    flops = 2.0 * n * (h * w * (npool * nf + 2 * nf * nf + 9 * (nf + c) * nf + nf * 9 * conv0_filters)
                       + oh * ow * ((conv0_filters + 3) * 9 * kb_filters + conv0_filters))
    tiles = n * (-(-oh // 8)) * (-(-ow // 16))
    executed = tiles * (27 * 6 + 22 * 12 + 36 * 9 + 8 * 15) * 2.0 * 16 * 16 * 32   # chain, 3x3 pairs, conv0, conv_depth: MFMAs of 16 x 16 x 32 per tile
    amin, amax = _int_array(mins), _int_array(maxs)
    status = _launch("s2d_depth_front", flops,
                     lambda: lib.kbn_s2d_depth_front_forward(xptr, xbs, kinv.data_ptr(), packed_s2d.data_ptr(), packed_weight.data_ptr(), optr, obs,
                                                             zptr, zbs, n, c, amin, len(mins), amax, len(maxs), 3, nf, conv0_filters, kb_filters,
                                                             h, w, float(s2d_negative_slope), float(conv0_negative_slope), float(kb_negative_slope),
                                                             0 if proj_negative_slope is None else 1,
                                                             0.0 if proj_negative_slope is None else float(proj_negative_slope),
                                                             _slot_ptr(out_depth_absmax, n), _stream()),
                     executed=executed, pipe="fp16", nbytes=4.0 * n * (h * w * c + oh * ow * (kb_filters + 3)))
    if status == _lib.KBN_ERR_UNSUPPORTED:
        if PROFILE is not None:
            PROFILE.pop()
        return None
    check(status, "kbn_s2d_depth_front_forward")
    return out_depth, xyz


# ------------------------------------------------------- pre-model stage / evaluation
@_on_tensor_device
def preprocess(image, sparse_depth, kernel_size: int = 7, threshold: float = 1.5, normalize_image: bool = True,
               normalized_image_range=(0, 1)):
    """Validity map + outlier removal (+ image normalisation): what reference src/kbnet.py:899-912 does
    before calling the model.  `normalized_image_range` as run_kbnet.py --normalized_image_range (reference
    src/transforms.py:185-214): [0, 1] image / 255, [-1, 1] 2 (image / 255) - 1, [0, 255] the image as it came (the reference returns
    its input for that range), anything else ValueError.  Returns (image, filtered_validity, filtered_sparse); image is None only
    with normalize_image=False (the caller keeps its own tensor) or when none was passed."""
    rng = [float(v) for v in normalized_image_range]
    untouched = rng == [0.0, 255.0]
    if untouched:
        normalize_image = False
    elif rng not in ([0.0, 1.0], [-1.0, 1.0]):
        raise ValueError("Unsupported normalization range: {}".format(list(normalized_image_range)))
    lib = _lib.load()
    _require(sparse_depth, "sparse_depth", 4)
    sd = sparse_depth.contiguous()
    n, _, h, w = sd.shape
    img = out_img = None
    c = 0
    if image is not None and normalize_image:
        _require(image, "image", 4)
        img = image.contiguous()
        c = img.shape[1]
        out_img = torch.empty_like(img)
    validity = torch.empty_like(sd)
    filtered = torch.empty_like(sd)
    ws = torch.empty(1, device=sd.device, dtype=torch.int32)
    check(lib.kbn_preprocess_forward(img.data_ptr() if img is not None else None, sd.data_ptr(),
                                     out_img.data_ptr() if out_img is not None else None, validity.data_ptr(),
                                     filtered.data_ptr(), ws.data_ptr(), 4, n, c, h, w, int(kernel_size),
                                     float(threshold), 1 if rng == [-1.0, 1.0] else 0, _stream()), "kbn_preprocess_forward")
    if untouched and image is not None:
        out_img = image.contiguous()
    return out_img, validity, filtered


@_on_tensor_device
def unpack_frames(image_u8, depth_raw, width: Optional[int] = None, x_offset: int = 0):
    """Decoded PNG pixels on the device -> (image N x 3 x H x W float32 in 0..255, sparse depth
    N x 1 x H x W float32 = raw / 256): the tensors reference src/datasets.py:259-283 returns.
    image_u8: N x H x Wraw x C uint8 (C in 1, 3, 4) or None; `width` / `x_offset` select the columns (the
    middle third of a triplet: width = Wraw // 3, x_offset = width).  depth_raw: N x H x W int16 (bit
    pattern of the 16-bit PNG samples) or uint8, or None."""
    lib = _lib.load()
    if image_u8 is None and depth_raw is None:
        raise KbnError("unpack_frames: nothing to unpack")
    image = depth = None
    n = h = wraw = c = 0
    if image_u8 is not None:
        if not image_u8.is_cuda or image_u8.dtype != torch.uint8 or image_u8.dim() != 4:
            raise KbnError("image_u8: expected a CUDA/HIP uint8 tensor N x H x W x C")
        image_u8 = image_u8.contiguous()
        n, h, wraw, c = image_u8.shape
        if width is None:
            width = wraw
        image = torch.empty((n, 3, h, width), device=image_u8.device, dtype=torch.float32)
    bits = 16
    if depth_raw is not None:
        if not depth_raw.is_cuda or depth_raw.dim() != 3 or depth_raw.dtype not in (torch.int16, torch.uint8):
            raise KbnError("depth_raw: expected a CUDA/HIP int16 (16-bit samples) or uint8 tensor N x H x W")
        depth_raw = depth_raw.contiguous()
        bits = 16 if depth_raw.dtype == torch.int16 else 8
        if image_u8 is not None and (depth_raw.shape[0] != n or depth_raw.shape[1] != h or depth_raw.shape[2] != width):
            raise KbnError(f"depth_raw has shape {tuple(depth_raw.shape)}, expected {(n, h, width)}")
        n, h, width = depth_raw.shape
        depth = torch.empty((n, 1, h, width), device=depth_raw.device, dtype=torch.float32)
    check(lib.kbn_unpack_frames_forward(image_u8.data_ptr() if image_u8 is not None else None,
                                        depth_raw.data_ptr() if depth_raw is not None else None,
                                        image.data_ptr() if image is not None else None,
                                        depth.data_ptr() if depth is not None else None,
                                        n, h, width, wraw if image_u8 is not None else width, int(x_offset),
                                        c if image_u8 is not None else 3, bits, _stream()), "kbn_unpack_frames_forward")
    return image, depth


@_on_tensor_device
def eval_metrics(output_depth, ground_truth, ground_truth_validity, min_evaluate_depth: float,
                 max_evaluate_depth: float):
    """Per-frame (MAE [mm], RMSE [mm], iMAE [1/km], iRMSE [1/km]) as an N x 4 fp64 tensor, computed on
    the device (reference src/kbnet.py:932-950 does this on the host after a D2H copy)."""
    lib = _lib.load()
    o = output_depth.contiguous()
    _require(o, "output_depth")
    g = ground_truth.contiguous()
    v = ground_truth_validity.contiguous()
    _require(g, "ground_truth")
    _require(v, "ground_truth_validity")
    n, h, w = o.shape[0], o.shape[-2], o.shape[-1]
    if g.numel() != o.numel() or v.numel() != o.numel():
        raise KbnError("ground truth must have one value per output pixel")
    sums = torch.zeros((n, 5), device=o.device, dtype=torch.float64)
    check(lib.kbn_eval_accumulate(o.data_ptr(), g.data_ptr(), v.data_ptr(), sums.data_ptr(), n, h, w,
                                  float(min_evaluate_depth), float(max_evaluate_depth), _stream()),
          "kbn_eval_accumulate")
    cnt = sums[:, 4:5].clamp_min(1.0)
    mean = sums[:, :4] / cnt
    return torch.stack([mean[:, 0], mean[:, 1].sqrt(), mean[:, 2], mean[:, 3].sqrt()], dim=1)
