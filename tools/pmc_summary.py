#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (last step's launches).
usage: pmc_summary.py <dir-with-*_counter_collection.csv> [launches-per-step]"""
import csv, glob, sys, collections, os
d = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    rows += list(csv.DictReader(open(f)))
# dispatch -> {counter: value}, keep kernel name & duration; dispatch ids restart per run, so key on (file-run) via counters
by = collections.OrderedDict()
for r in rows:
    if "kbn::" not in r["Kernel_Name"] or "pack_weight" in r["Kernel_Name"] or "intrinsics" in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"], r["Grid_Size"], r["LDS_Block_Size"], int(r["Dispatch_Id"]))
    e = by.setdefault(key, {})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
    e.setdefault("_dur", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
# keep the last occurrence of each (name, grid, lds) signature sequence: take final third of dispatches
keys = list(by.keys())
maxd = max(k[3] for k in keys)
sel = [k for k in keys if k[3] > maxd - int(sys.argv[2]) if len(sys.argv) > 2] if len(sys.argv) > 2 else keys[-32:]
print(f"{'kernel':34s} {'grid':>8s} {'us':>7s} {'mfma%':>6s} {'act%':>6s} {'wait%':>6s} {'winst%':>6s} {'ldsbc%':>6s} {'fetchMB':>8s} {'writeMB':>8s} {'valu/mfma':>9s} {'lds/mfma':>8s}")
for k in sel:
    e = by[k]
    nm = k[0].replace("void kbn::", "").replace("kbn::", "").split("(")[0][:34]
    wc = e.get("SQ_WAVE_CYCLES", 0) or 1
    busy = e.get("SQ_BUSY_CYCLES", 0) or 1
    g = lambda n: e.get(n, float("nan"))
    dur = sum(e["_dur"]) / len(e["_dur"])
    mf = g("SQ_INSTS_MFMA") or float("nan")
    print(f"{nm:34s} {k[1]:>8s} {dur:7.1f} {100*g('SQ_VALU_MFMA_BUSY_CYCLES')/(g('GRBM_GUI_ACTIVE')/8*1024) if 'GRBM_GUI_ACTIVE' in e else 100*g('SQ_VALU_MFMA_BUSY_CYCLES')/busy/4:6.1f} "
          f"{100*g('SQ_ACTIVE_INST_ANY')/wc:6.1f} {100*g('SQ_WAIT_ANY')/wc:6.1f} {100*g('SQ_WAIT_INST_ANY')/wc:6.1f} "
          f"{100*g('SQ_LDS_BANK_CONFLICT')/(g('SQ_LDS_IDX_ACTIVE') or 1):6.1f} {2*g('FETCH_SIZE')/1024:8.1f} {g('WRITE_SIZE')/1024:8.1f} "
          f"{g('SQ_INSTS_VALU')/mf:9.2f} {g('SQ_INSTS_LDS')/mf:8.2f}")
