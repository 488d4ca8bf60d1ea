"""Mutated PNG files through the library reader: truncations, flipped bytes, IHDR fields, chunk lengths, huge dimensions, trailing bytes.
Every call must come back with a status -- run by tests/test_loader.py in a process of its own, so that a crash is a failed test.
usage: png_fuzz.py [iterations]"""
import sys, os, glob, random, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kbnet_amd as kb
lib = kb._lib.load()
files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io", "*.png")))
r = random.Random(5)
n_ok = n_err = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3000):
    data = bytearray(open(r.choice(files), "rb").read())
    kind = r.randrange(6)
    if kind == 0:
        data = data[:r.randrange(0, len(data))]
    elif kind == 1:
        for _ in range(r.randrange(1, 8)):
            data[r.randrange(len(data))] = r.randrange(256)
    elif kind == 2:   # IHDR fields
        off = 16 + r.randrange(13)
        data[off] = r.randrange(256)
    elif kind == 3:   # a chunk length
        pos = 8
        chunks = []
        while pos + 8 <= len(data):
            ln = int.from_bytes(data[pos:pos + 4], "big"); chunks.append(pos); pos += 12 + ln
        p = r.choice(chunks)
        data[p:p + 4] = r.choice([0, 1, 0x7fffffff, 0xffffffff, len(data), r.randrange(1 << 32)]).to_bytes(4, "big")
    elif kind == 4:   # huge dimensions
        data[16:24] = r.choice([0, 1, 65535, 1 << 20, 0x7fffffff, 0xffffffff]).to_bytes(4, "big") + r.choice([0, 1, 65535, 1 << 20, 0x7fffffff]).to_bytes(4, "big")
    else:
        data += bytes(r.randrange(256) for _ in range(r.randrange(1, 64)))
    b = bytes(data)
    w, h, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib.kbn_png_info(b, len(b), C.byref(w), C.byref(h), C.byref(c), C.byref(d))
    if rc != 0:
        n_err += 1
        continue
    need = w.value * h.value * c.value * (2 if d.value == 16 else 1)
    size = min(need, 1 << 24) if r.random() < 0.8 else r.randrange(0, 4096)
    buf = (C.c_ubyte * max(size, 1))()
    rc = lib.kbn_png_decode(b, len(b), buf, size)
    n_ok += rc == 0
    n_err += rc != 0
print("survived", n_ok, "decoded,", n_err, "rejected")
