"""Builds csrc/*.hip into libkbnet_hip.so (gfx950) with hipcc, in-tree.

The shared library is git-ignored but travels to the GPU box with the snapshot.
`python -m kbnet_build` style use: `from kbnet_amd import _build; _build.build()`.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkbnet_hip.so")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")
# longest translation units first (conv_split.hip alone takes 100-110 s): the pool then ends with the short ones
SOURCES = ["conv_split.hip", "conv_igemm.hip", "conv_dma_t1.hip", "conv_dma_t2.hip", "conv_dma_t4.hip", "conv_up2x.hip", "front.hip", "kb_pair_nb3.hip",
           "kb_pair_nb4.hip", "conv_wino.hip", "s2d.hip", "kb_pair.hip", "head.hip", "tail.hip", "kb.hip", "conv_dma.hip", "tune.hip", "abi.hip", "pre_eval.hip",
           "unpack.hip", "io_png.hip", "elementwise.hip"]
HEADERS = ["kbn_common.h", "conv_common.h", "conv_dma_impl.h", "kb_pair_impl.h", "front_common.h", "s2d_pools.h", "s2d_stage.h", "conv_split_body.inc", "upconv64_split_body.inc", os.path.join("..", "..", "include", "kbnet_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall",
         "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (needed to build libkbnet_hip.so)")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    extra = os.environ.get("KBN_HIPCC_FLAGS", "").split()  # experiments, e.g. -DKBN_WAVES_PER_SIMD=3
    if extra:
        force = True
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + FLAGS + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[kbnet build]", " ".join(os.path.relpath(c) if os.path.isabs(c) else c for c in cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 4))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
