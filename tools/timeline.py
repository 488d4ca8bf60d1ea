#!/usr/bin/env python3
"""Per-launch timeline of the last bench step from a rocprofv3 kernel trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
ks = [r for r in rows if "kbn::" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]]
per = len(ks) // steps
last = ks[-per:]
t0 = int(last[0]["Start_Timestamp"])
tot = 0
for r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    nm = r["Kernel_Name"].replace("void kbn::", "").replace("kbn::", "").split("(")[0][:52]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {d:8.1f}us wgs={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):>6} vgpr={r.get('VGPR_Count')}+{r.get('Accum_VGPR_Count')} {nm}")
print("sum of kernel time: %.1f us; span %.1f us" % (tot, (int(last[-1]["End_Timestamp"]) - t0) / 1e3))
