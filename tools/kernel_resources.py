#!/usr/bin/env python3
"""Registers / LDS / scratch of every kernel in hipcc -save-temps .s files (what bounds its residency per CU).
usage: kernel_resources.py file.s [...]"""
import re, sys
for path in sys.argv[1:]:
    s = open(path).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        body = m.group(2)
        g = lambda k: int((re.search(r"\.amdhsa_" + k + r"\s+(\d+)", body) or [0, 0])[1])
        vg, acc = g("next_free_vgpr"), g("accum_offset")
        name = m.group(1)
        print(f"{name[:110]:110s} vgpr(arch+acc) {vg:4d} accum_offset {acc:4d} sgpr {g('next_free_sgpr'):4d} lds {g('group_segment_fixed_size'):6d} scratch {g('private_segment_fixed_size'):5d}")
