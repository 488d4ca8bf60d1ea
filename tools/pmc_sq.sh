#!/bin/bash
# SQ wait / LDS-conflict counters of every kernel of one eager forward pass (two --pmc passes, kernel-trace off):
#   tools/pmc_sq.sh <outfile>     -> per kernel: busy / wait fractions and LDS bank-conflict share
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/pmc_sq.txt}
case "$OUT" in /*) ;; *) OUT="$R/$OUT";; esac
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_sq/$tag -o p -- python $R/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-void --no-side-batch < /dev/null > /tmp/pmc_sq_$tag.log 2>&1 || tail -5 /tmp/pmc_sq_$tag.log
done
python - "$OUT" <<'PY'
import csv, glob, collections, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob("/tmp/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kbn::" not in r["Kernel_Name"]: continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][:70]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name].add(r["Dispatch_Id"])
with open(sys.argv[1], "w") as o:
    hdr = f"{'kernel':70s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} | {'lds_conflict/idx':>16s} {'wait_lds':>8s} | {'valu/mfma':>9s} {'lds/mfma':>8s}"
    print(hdr); o.write(hdr + "\n")
    for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        w = c.get("SQ_WAVE_CYCLES", 0) or 1.0
        idx = c.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0
        mf = c.get("SQ_INSTS_MFMA", 0) or 1.0
        line = (f"{name:70s} {c.get('SQ_WAIT_ANY',0)/w:8.3f} {c.get('SQ_WAIT_INST_ANY',0)/w:9.3f} {c.get('SQ_ACTIVE_INST_ANY',0)/w:7.3f} | "
                f"{c.get('SQ_LDS_BANK_CONFLICT',0)/idx:16.3f} {c.get('SQ_WAIT_INST_LDS',0)/w:8.3f} | {(c.get('SQ_INSTS_VALU',0)-c.get('SQ_INSTS_MFMA',0))/mf:9.2f} {c.get('SQ_INSTS_LDS',0)/mf:8.2f}")
        print(line); o.write(line + "\n")
PY
