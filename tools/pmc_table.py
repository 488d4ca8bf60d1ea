#!/usr/bin/env python3
"""Per-kernel table from a directory of rocprofv3 --pmc passes (tools/pmc_cmd.sh): average per dispatch of
duration, MFMA-busy share, vector / LDS instructions per MFMA, LDS bank-conflict share, wait shares.
usage: pmc_table.py <dir> [min_us]"""
import csv, glob, collections, os, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        if "kbn::" not in r["Kernel_Name"]:
            continue
        k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("kbn::", "").replace("(anonymous namespace)::", "").split("(")[0]
        k = k + " g" + r["Grid_Size"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Counter_Name"], r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Counter_Name"], r["Dispatch_Id"])); n[k][r["Counter_Name"]] += 1
            if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
print(f"{'kernel':58s} {'n':>3s} {'us':>8s} {'mfma%':>6s} {'valu/mfma':>9s} {'lds/mfma':>8s} {'salu/mfma':>9s} {'ldsbc%':>6s} {'wait%':>6s} {'winst%':>6s}")
rows = []
for k, e in acc.items():
    g = lambda c: e.get(c, 0.0) / max(n[k].get(c, 1), 1)
    d = sum(dur[k]) / max(len(dur[k]), 1)
    if d < min_us:
        continue
    mf = g("SQ_INSTS_MFMA") or float("nan")
    busy = 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 1024) if g("GRBM_GUI_ACTIVE") else float("nan")
    wc = g("SQ_WAVE_CYCLES") or 1
    rows.append((d * len(dur[k]), f"{k[:58]:58s} {len(dur[k]):3d} {d:8.1f} {busy:6.1f} {g('SQ_INSTS_VALU') / mf:9.2f} {g('SQ_INSTS_LDS') / mf:8.2f} "
                 f"{g('SQ_INSTS_SALU') / mf:9.2f} {100 * g('SQ_LDS_BANK_CONFLICT') / (g('SQ_LDS_IDX_ACTIVE') or 1):6.1f} "
                 f"{100 * g('SQ_WAIT_ANY') / wc:6.1f} {100 * g('SQ_WAIT_INST_ANY') / wc:6.1f}"))
for _, line in sorted(rows, reverse=True):
    print(line)
