#!/usr/bin/env python3
"""Compact instruction-class view of one kernel in a hipcc -save-temps .s file.
usage: isa_view.py file.s <mangled-name-regex>"""
import re, sys, textwrap
s = open(sys.argv[1]).read()
m = re.search(r"^(" + sys.argv[2] + r"):[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S | re.M)
body = m.group(2).split("\n")
out = []
for l in body:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        if l.startswith(".LBB"):
            out.append(" " + l + " ")
        continue
    op = l.split()[0]
    if op.startswith("v_mfma"): k = "M"
    elif op.startswith("global_load_lds"): k = "D"
    elif op.startswith("global_load") or op.startswith("buffer_load"): k = "G"
    elif op.startswith("ds_read"): k = "r"
    elif op.startswith("ds_write"): k = "w"
    elif op.startswith("s_waitcnt"): k = " [" + l.split(None, 1)[1] + "] "
    elif op.startswith("s_barrier"): k = " |BAR| "
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): k = " <" + " ".join(l.split()[:2]) + "> "
    elif op.startswith("global_store"): k = "S"
    elif op.startswith("s_load"): k = "L"
    elif op.startswith("s_"): k = ","
    else: k = "."
    out.append(k)
print(len(body), "lines")
print(textwrap.fill("".join(out), 180))
