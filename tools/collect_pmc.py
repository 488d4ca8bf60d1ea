#!/usr/bin/env python3
"""Per-kernel PMC evidence for bench.py's workload (KITTI 352x1216, batch 32 unless given, eager whole-batch launches):
  pass 1  --pmc FETCH_SIZE                      HBM read KiB  (gfx950: counts a 128-byte request as 64 B -> doubled)
  pass 2  --pmc WRITE_SIZE                      HBM write KiB
  pass 3  --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
          matrix-pipe utilisation = MFMA busy cycles (summed over the chip's 1024 SIMDs)
                                    / (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 * 1024 SIMDs)
Separate passes as MI355X_MICROARCH.md prescribes (TCC slots), no tracing flags next to --pmc.  First-use tuning is
replayed from the newest profiles/r*/v*_tune_cache.txt (KBN_TUNE_CACHE) so that no timing launches are averaged in,
and the VOID side measurement is skipped: every launch counted is one of the bench's whole-batch KITTI launches.
usage (GPU box): python tools/collect_pmc.py <out.json> [frames_per_gpu]"""
import csv, glob, json, os, subprocess, sys, collections, re
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_json = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tmp = "/tmp/kbn_pmc"
os.makedirs(tmp, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
caches = sorted(glob.glob(os.path.join(root, "profiles", "r*", "v*_tune_cache.txt")),
                key=lambda p: int(re.search(r"v(\d+)_", os.path.basename(p)).group(1)))
if caches and "KBN_TUNE_CACHE" not in env:
    env["KBN_TUNE_CACHE"] = "/tmp/kbn_pmc_tune_cache.txt"      # a copy: the run may append shapes
    open(env["KBN_TUNE_CACHE"], "w").write(open(caches[-1]).read())
PASSES = {"FETCH_SIZE": ["FETCH_SIZE"], "WRITE_SIZE": ["WRITE_SIZE"],
          "MFMA": ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE"]}
res = collections.defaultdict(lambda: collections.defaultdict(float))
for tag, counters in PASSES.items():
    subprocess.run(["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", tmp, "-o", tag, "--",
                    sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--eager",
                    "--no-cpu-baseline", "--no-void", "--no-side-batch", "--frames-per-gpu", str(frames)], cwd="/tmp", env=env, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, stdin=subprocess.DEVNULL)
    f = glob.glob(os.path.join(tmp, "**", tag + "_counter_collection.csv"), recursive=True)[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        if "kbn::" not in r["Kernel_Name"]:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0]
        res[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if (name, r["Dispatch_Id"]) not in seen:
            seen.add((name, r["Dispatch_Id"]))
            res[name]["launches_" + tag] += 1
            res[name]["ns_" + tag] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
final = {}
for name, e in res.items():
    n = e["launches_FETCH_SIZE"]
    if not n:
        continue
    fetch = 2.0 * e["FETCH_SIZE"] * 1024 / n                   # gfx950: FETCH_SIZE reads 1/2 of a wide stream
    write = e["WRITE_SIZE"] * 1024 / max(e["launches_WRITE_SIZE"], 1)
    row = {"hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "launches": int(n),
           "avg_us_under_pmc": e["ns_FETCH_SIZE"] / n / 1e3}
    if e["GRBM_GUI_ACTIVE"] > 0 and e["SQ_INSTS_MFMA"] > 0:
        row["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        row["mfma_insts_per_launch"] = e["SQ_INSTS_MFMA"] / e["launches_MFMA"]
        row["shader_clock_ghz_under_pmc"] = e["GRBM_GUI_ACTIVE"] / 8.0 / e["ns_MFMA"]
    final[name] = row
json.dump({"note": "rocprofv3 --pmc, three separate passes (FETCH_SIZE x2 on gfx950 | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES "
                   "SQ_INSTS_MFMA GRBM_GUI_ACTIVE), averaged over the whole-batch launches of "
                   f"bench.py --eager --steps 2 --warmup 1 --no-void --frames-per-gpu {frames} with the tuner's choices replayed; "
                   "mfma_busy_frac = MFMA busy cycles / (GUI_ACTIVE per XCD x 1024 SIMDs)",
           "frames_per_gpu": frames, "kernels": final}, open(out_json, "w"), indent=1)
print(json.dumps({k: [round(v["hbm_bytes_per_launch"] / 1e6, 1), round(v.get("mfma_busy_frac", 0), 3)] for k, v in final.items()}, indent=1))
