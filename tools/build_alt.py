#!/usr/bin/env python3
"""An alternative build of the library for same-box A/Bs (tools/ab_lib.sh, KBN_LIB_PATH): the named sources recompiled with extra hipcc flags,
every other object taken from the shipped build.   usage: build_alt.py <name> <flags, e.g. -DKBN_S2D_NO_ROT=1> <source.hip> [...]  ->  alt/<name>.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
b = importlib.import_module("kbnet_amd._build")
b.build(force=False, verbose=False)
name, flags, srcs = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
out_dir = os.path.join(ROOT, "alt")
os.makedirs(os.path.join(out_dir, "obj_" + name), exist_ok=True)
objs = []
for s in b.SOURCES:
    o = os.path.join(b.OBJ_DIR, s.replace(".hip", ".o"))
    if s in srcs:
        o = os.path.join(out_dir, "obj_" + name, s.replace(".hip", ".o"))
        subprocess.run([b._hipcc()] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, s), "-o", o], check=True)
    objs.append(o)
lib = os.path.join(out_dir, name + ".so")
subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lz"], check=True)
print(lib)
