#!/usr/bin/env python3
"""Runs bench.py under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, as
MI355X_MICROARCH.md prescribes) and writes per-kernel HBM bytes per launch to a JSON file.
gfx950 correction: FETCH_SIZE counts 128-byte requests as 64 B -> doubled.  Units: KiB.
usage (GPU box): python tools/collect_traffic.py <out.json>"""
import csv, glob, json, os, subprocess, sys, collections, re
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_json = sys.argv[1]
tmp = "/tmp/kbn_traffic"
os.makedirs(tmp, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
res = collections.defaultdict(lambda: {"fetch_kib": 0.0, "write_kib": 0.0, "launches": 0})
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", counter, "--",
                    sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--eager",
                    "--no-cpu-baseline"], cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(tmp, "**", counter + "_counter_collection.csv"), recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if "kbn::" not in r["Kernel_Name"] or r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        e = res[name]
        e["fetch_kib" if counter == "FETCH_SIZE" else "write_kib"] += float(r["Counter_Value"])
        if counter == "FETCH_SIZE":
            e["launches"] += 1
final = {}
for name, e in res.items():
    if not e["launches"]:
        continue
    fetch = 2.0 * e["fetch_kib"] * 1024 / e["launches"]   # gfx950: FETCH_SIZE reads 1/2 of a wide stream
    write = e["write_kib"] * 1024 / e["launches"]
    final[name] = {"hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "launches": e["launches"]}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 (gfx950), averaged over launches of bench.py --eager --steps 2 --warmup 1",
           "kernels": final}, open(out_json, "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in final.items()}, indent=1))
