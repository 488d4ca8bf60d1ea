#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats (default launch and --branches 1) and the PMC
# collection, all with ONE set of first-use tuning choices (KBN_TUNE_CACHE) so that no timing launches pollute the
# statistics.  usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
set -u
TAG=${1:-vX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
export KBN_TUNE_CACHE=$O/${TAG}_tune_cache.txt
rm -f $KBN_TUNE_CACHE
cd /tmp; export TMPDIR=/tmp
timeout 600 python $R/bench.py < /dev/null > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
timeout 300 python $R/bench.py --branches 1 --no-cpu-baseline --no-void --no-side-batch --no-fp32-mfma --no-bf16 --no-fp16 --no-mixed --no-sustained --no-batch1 --no-options < /dev/null > /dev/null 2>&1   # whole-batch shapes into the cache
for mode in "" "--branches 1"; do
  sfx=$( [ -z "$mode" ] && echo "" || echo "_branches1" )
  # the whole-batch pass is the per-launch table: one kernel at a time (no level side branches in its single-branch graph)
  rm -rf /tmp/kbn_prof; KBN_NO_OVERLAP=$( [ -z "$mode" ] && echo 0 || echo 1 ) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kbn_prof -o p -- \
      python $R/bench.py $mode --no-cpu-baseline --no-void --no-side-batch --no-fp32-mfma --no-bf16 --no-fp16 --no-mixed --no-sustained --no-batch1 --no-options < /dev/null > $O/${TAG}_prof$sfx.log 2>&1
  cp $(find /tmp/kbn_prof -name "p_kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats$sfx.csv
done
timeout 900 python $R/tools/collect_pmc.py $O/${TAG}_traffic.json < /dev/null > $O/${TAG}_pmc.log 2>&1
tail -3 $O/${TAG}_pmc.log
head -c 600 $O/${TAG}_bench.json; echo
head -6 $O/${TAG}_kernel_stats_branches1.csv | cut -c1-150
