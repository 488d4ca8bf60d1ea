#!/usr/bin/env python3
"""Instruction-class mix of the hottest basic blocks (most MFMAs) of every kernel in a hipcc -save-temps .s file.
usage: isa_mix.py file.s [name-substring]"""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", s, re.S | re.M):
    if pat not in m.group(1):
        continue
    blocks, cur, name = [], [], "entry"
    for l in m.group(0).split("\n"):
        if re.match(r"\.LBB\d+_\d+:", l.strip()):
            blocks.append((name, cur)); cur = []; name = l.strip().split(":")[0]
        else:
            cur.append(l)
    blocks.append((name, cur))
    blocks.sort(key=lambda b: -sum("v_mfma" in x for x in b[1]))
    print(m.group(1)[:90])
    for name, body in blocks[:2]:
        c = collections.Counter()
        for l in body:
            t = l.strip().split()
            if not t or t[0].startswith((";", ".")):
                continue
            o = t[0]
            k = ("mfma" if o.startswith("v_mfma") else "valu" if o.startswith("v_") else "wait" if o.startswith("s_waitcnt")
                 else "salu" if o.startswith("s_") else "lds" if o.startswith("ds_") else "vmem" if o.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
            c[k] += 1
            if k == "valu":
                c["v:" + o] += 1
        if not c["mfma"]:
            continue
        print("   ", name, {k: v for k, v in c.items() if not k.startswith("v:")}, {k[2:]: v for k, v in c.items() if k.startswith("v:")})
