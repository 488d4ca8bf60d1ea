"""Input pipeline: PNG / .npy files -> batched device tensors.

The reference feeds inference from `datasets.KBNetInferenceDataset` through a one-worker, batch-1
`torch.utils.data.DataLoader` and moves every sample to the device on its own
(reference src/kbnet.py:764-772, 887-896; src/datasets.py:229-286; PIL decode in
src/data_utils.py:58-152).  At > 1000 frames/s per GPU that loader is the bottleneck, so here

  * a pool of host threads decodes the PNG files with the library's own reader
    (`kbn_png_decode`: zlib inflate + scanline filters in C++; ctypes drops the GIL around the call)
    straight into PINNED staging tensors, a whole batch at a time;
  * the raw bytes (3 B/px image, 2 B/px depth instead of 12 + 4 B/px of float32) cross PCIe with one
    asynchronous copy per tensor on a side stream, overlapped with the previous batch's forward;
  * `kbn_unpack_frames_forward` turns them into the dataset's tensors on the device (triplet crop,
    uint8 -> float32, depth / 256); `ops.preprocess` (row f1) continues from there.

`load_image`, `load_depth`, `load_image_triplet` mirror the reference functions of the same names
(numpy in, numpy out) on top of the same decoder.
"""

from __future__ import annotations

import ctypes as C
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import KbnError, check


# ------------------------------------------------------------------------- PNG decode
def png_info(data: bytes) -> Tuple[int, int, int, int]:
    """(width, height, channels, bit_depth) of a PNG file held in memory."""
    lib = _lib.load()
    w, h, c, b = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(lib.kbn_png_info(data, len(data), C.byref(w), C.byref(h), C.byref(c), C.byref(b)), "kbn_png_info")
    return w.value, h.value, c.value, b.value


def decode_png(data: bytes, out: Optional[np.ndarray] = None) -> np.ndarray:
    """PNG bytes -> H x W x C uint8 (H x W for gray) or H x W uint16 array; decodes into `out` if given
    (any C-contiguous buffer of the right size, e.g. the numpy view of a pinned tensor)."""
    lib = _lib.load()
    w, h, c, b = png_info(data)
    dtype = np.uint16 if b == 16 else np.uint8
    shape = (h, w) if c == 1 else (h, w, c)
    if out is None:
        out = np.empty(shape, dtype=dtype)
    if out.nbytes < h * w * c * dtype().itemsize or not out.flags["C_CONTIGUOUS"]:
        raise KbnError("decode_png: output buffer too small or not contiguous")
    check(lib.kbn_png_decode(data, len(data), out.ctypes.data_as(C.c_void_p), out.nbytes), "kbn_png_decode")
    return out


def decode_png_batch(files: Sequence[bytes], outs: Sequence[np.ndarray], threads: int = 8) -> None:
    """Decodes files[i] into outs[i] (C-contiguous numpy buffers of the right size) on `threads` host
    threads inside the library (`kbn_png_decode_batch`); raises on the first damaged file."""
    lib = _lib.load()
    n = len(files)
    if n != len(outs):
        raise KbnError("decode_png_batch: one output buffer per file")
    for o in outs:
        if not o.flags["C_CONTIGUOUS"]:
            raise KbnError("decode_png_batch: output buffers must be contiguous")
    fptr = (C.c_char_p * n)(*files)
    flen = (C.c_size_t * n)(*[len(f) for f in files])
    optr = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    olen = (C.c_size_t * n)(*[o.nbytes for o in outs])
    status = (C.c_int * n)()
    check(lib.kbn_png_decode_batch(fptr, flen, optr, olen, n, int(threads), status), "kbn_png_decode_batch")


def _read(path: str) -> bytes:
    with open(path, "rb") as f:
        return f.read()


# ---------------------------------------------- the reference's loader functions (numpy)
def load_image(path: str, normalize: bool = True, data_format: str = "HWC") -> np.ndarray:
    """reference src/data_utils.py:58-85 (`Image.open(path).convert('RGB')` -> float32)."""
    px = decode_png(_read(path))
    if px.dtype != np.uint8:
        raise KbnError("load_image: 16-bit images are not supported")
    if px.ndim == 2:
        px = np.repeat(px[:, :, None], 3, axis=2)
    image = np.asarray(px[:, :, :3], np.float32)
    if data_format == "CHW":
        image = np.transpose(image, (2, 0, 1))
    elif data_format != "HWC":
        raise ValueError("Unsupported data format: {}".format(data_format))
    return image / 255.0 if normalize else image


def load_image_triplet(path: str, normalize: bool = True):
    """reference src/datasets.py:22-46: images at t-1, t, t+1 (C x H x W each) -> (t, t-1, t+1) order."""
    images = load_image(path, normalize=normalize, data_format="CHW")
    image1, image0, image2 = np.split(images, indices_or_sections=3, axis=-1)
    return image1, image0, image2


def load_depth(path: str, data_format: str = "HW") -> np.ndarray:
    """reference src/data_utils.py:123-152: 16-bit PNG / 256, non-positive values -> 0."""
    z = np.array(decode_png(_read(path)), dtype=np.float32)
    if z.ndim != 2:
        raise KbnError("load_depth: expected a single-channel PNG")
    z = z / 256.0
    z[z <= 0] = 0.0
    if data_format == "HW":
        pass
    elif data_format == "CHW":
        z = np.expand_dims(z, axis=0)
    elif data_format == "HWC":
        z = np.expand_dims(z, axis=-1)
    else:
        raise ValueError("Unsupported data format: {}".format(data_format))
    return z


# ---------------------------------------------------------------------- output side
def encode_depth_png(samples: np.ndarray, level: int = 6) -> bytes:
    """H x W uint16 samples (depth * 256) -> the bytes of a 16-bit grayscale PNG (kbn_png_encode_gray16)."""
    lib = _lib.load()
    a = np.ascontiguousarray(samples, dtype=np.uint16)
    if a.ndim != 2:
        raise KbnError("encode_depth_png: expected an H x W array")
    h, w = a.shape
    cap = lib.kbn_png_encode_gray16_bound(w, h)
    buf = (C.c_ubyte * cap)()
    n = C.c_size_t()
    check(lib.kbn_png_encode_gray16(a.ctypes.data_as(C.c_void_p), w, h, buf, cap, C.byref(n), int(level)), "kbn_png_encode_gray16")
    return bytes(memoryview(buf)[:n.value])


def depth_samples(z) -> np.ndarray:
    """What reference src/data_utils.py:165-167 stores for a depth map: np.uint32(z * 256.0), clipped to the PNG's 16 bits as PIL
    clips a mode 'I' image -- as uint16.  `z`: a numpy array (converted here) or a device tensor of any shape
    (kbn_depth_to_u16_forward, then one device-to-host copy of 2 bytes per pixel)."""
    if torch.is_tensor(z) and z.is_cuda:
        zc = z.detach().contiguous().float()
        out = torch.empty(zc.shape, device=zc.device, dtype=torch.int16)
        with torch.cuda.device(zc.device):
            check(_lib.load().kbn_depth_to_u16_forward(zc.data_ptr(), out.data_ptr(), zc.numel(),
                                                       torch.cuda.current_stream().cuda_stream), "kbn_depth_to_u16_forward")
        return out.cpu().numpy().view(np.uint16)
    a = np.asarray(z.detach().cpu().numpy() if torch.is_tensor(z) else z, dtype=np.float32) * np.float32(256.0)
    return np.clip(np.nan_to_num(a, nan=0.0), 0.0, 65535.0).astype(np.uint32).astype(np.uint16)


def save_depth(z, path: str) -> None:
    """reference src/data_utils.py:154-167: a depth map (H x W, numpy or tensor) as a 16-bit PNG holding depth * 256."""
    s = depth_samples(z)
    s = s.reshape(s.shape[-2:]) if s.ndim > 2 and int(np.prod(s.shape[:-2])) == 1 else s
    with open(path, "wb") as f:
        f.write(encode_depth_png(s))


def save_depth_batch(z: torch.Tensor, paths: Sequence[str], threads: int = 8) -> None:
    """N x 1 x H x W (or N x H x W) depth maps -> one PNG each (run_kbnet.py --save_outputs, reference src/kbnet.py:1018-1026): ONE
    device pass and copy for the batch, the files encoded and written by `threads` host threads (the encoder runs without the GIL)."""
    s = depth_samples(z)
    s = s.reshape((-1,) + s.shape[-2:])
    if s.shape[0] != len(paths):
        raise KbnError(f"save_depth_batch: {s.shape[0]} maps, {len(paths)} paths")

    def one(i):
        with open(paths[i], "wb") as f:
            f.write(encode_depth_png(s[i]))

    with ThreadPoolExecutor(max_workers=max(1, int(threads))) as ex:
        list(ex.map(one, range(len(paths))))


# ------------------------------------------------------------------- batched device loader
class InferenceFrameLoader:
    """Iterates `(image N x 3 x H x W in 0..255, sparse_depth N x 1 x H x W, intrinsics N x 3 x 3)` device
    batches over the files `datasets.KBNetInferenceDataset` would read (same constructor arguments,
    reference src/datasets.py:229-257), in order, the last batch possibly short.  All frames must have
    one size.  Batch i+1 is decoded and copied while the caller works on batch i."""

    def __init__(self, image_paths: Sequence[str], sparse_depth_paths: Sequence[str],
                 intrinsics_paths: Sequence[str], use_image_triplet: bool = True, batch_size: int = 8,
                 device: Optional[torch.device] = None, workers: int = 8, prefetch: int = 4):
        self.n_sample = len(image_paths)
        for paths in (sparse_depth_paths, intrinsics_paths):
            assert len(paths) == self.n_sample
        if device is None or torch.device(device).type != "cuda":
            raise KbnError("InferenceFrameLoader: a CUDA/HIP device is required (no CPU fallback)")
        self.image_paths, self.sparse_depth_paths = list(image_paths), list(sparse_depth_paths)
        self.intrinsics_paths = list(intrinsics_paths)
        self.use_image_triplet = use_image_triplet
        self.batch_size = int(batch_size)
        self.device = torch.device(device)
        self.workers = max(1, int(workers))                      # decode threads per batch (inside the library)
        self.pool = ThreadPoolExecutor(max_workers=min(16, self.workers))   # file reads
        self.copy_stream = torch.cuda.Stream(device=self.device)
        # `prefetch` batches are decoded concurrently (one PNG is one serial inflate stream, so the decode
        # parallelism is batch_size x prefetch), each in its own set of pinned staging tensors
        self.prefetch = max(1, int(prefetch))
        self._driver = ThreadPoolExecutor(max_workers=self.prefetch)
        self._stage: List[Optional[dict]] = [None] * (self.prefetch + 1)

    def __len__(self):
        return (self.n_sample + self.batch_size - 1) // self.batch_size

    # -- one batch: decode on the pool into pinned memory, async H2D + unpack on the copy stream --
    def _staging(self, slot: int, hraw: int, wraw: int, c: int, hd: int, wd: int, depth_dtype):
        st = self._stage[slot]
        key = (hraw, wraw, c, hd, wd, depth_dtype)
        if st is None or st["key"] != key:
            st = {"key": key,
                  "image": torch.empty((self.batch_size, hraw, wraw, c), dtype=torch.uint8).pin_memory(),
                  "depth": torch.empty((self.batch_size, hd, wd), dtype=depth_dtype).pin_memory(),
                  "k": torch.empty((self.batch_size, 3, 3), dtype=torch.float32).pin_memory(),
                  "done": None}
            self._stage[slot] = st
        return st

    def _submit(self, b: int):
        lo, hi = b * self.batch_size, min((b + 1) * self.batch_size, self.n_sample)
        idx = list(range(lo, hi))
        files = list(self.pool.map(_read, [self.image_paths[i] for i in idx] + [self.sparse_depth_paths[i] for i in idx]))
        img_files, dep_files = files[:len(idx)], files[len(idx):]
        wraw, hraw, c, bits = png_info(img_files[0])
        wd, hd, cd, dbits = png_info(dep_files[0])
        if bits != 8 or cd != 1:
            raise KbnError("InferenceFrameLoader: expected 8-bit images and single-channel depth PNGs")
        width = wraw // 3 if self.use_image_triplet else wraw
        if (hd, wd) != (hraw, width):
            raise KbnError(f"depth map is {hd} x {wd}, image is {hraw} x {width}")
        st = self._staging(b % (self.prefetch + 1), hraw, wraw, c, hd, wd, torch.int16 if dbits == 16 else torch.uint8)
        if st["done"] is not None:
            st["done"].synchronize()      # the copies that last used this staging set
        img_np, dep_np, k_np = st["image"].numpy(), st["depth"].numpy(), st["k"].numpy()
        dep_view = dep_np.view(np.uint16) if dbits == 16 else dep_np

        nb = len(idx)
        for j in range(nb):
            if png_info(img_files[j]) != (wraw, hraw, c, 8) or png_info(dep_files[j]) != (wd, hd, 1, dbits):
                raise KbnError("InferenceFrameLoader: all frames of a run must have one size and format")
            k_np[j] = np.load(self.intrinsics_paths[idx[j]]).astype(np.float32)
        # one library call decodes the whole batch on its own threads (images first: they are the long poles)
        decode_png_batch(img_files + dep_files, [img_np[j] for j in range(nb)] + [dep_view[j] for j in range(nb)],
                         threads=self.workers)
        with torch.cuda.stream(self.copy_stream):
            img_d = st["image"][:nb].to(self.device, non_blocking=True)
            dep_d = st["depth"][:nb].to(self.device, non_blocking=True)
            k_d = st["k"][:nb].to(self.device, non_blocking=True)
            image, depth = ops.unpack_frames(img_d, dep_d, width=width, x_offset=width if self.use_image_triplet else 0)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        st["done"] = done
        return image, depth, k_d, done

    def __iter__(self):
        nb = len(self)
        pending = [self._driver.submit(self._submit, b) for b in range(min(self.prefetch, nb))]
        for b in range(nb):
            image, depth, k, done = pending.pop(0).result()
            if b + self.prefetch < nb:
                pending.append(self._driver.submit(self._submit, b + self.prefetch))
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)
            for t in (image, depth, k):
                t.record_stream(cur)
            yield image, depth, k
