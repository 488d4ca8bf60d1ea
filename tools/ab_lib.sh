#!/bin/bash
# same-box A/B of BUILDS of the library (compile-time variants): tools/ab_lib.sh "<alt1.so> [alt2.so ...]" [repeats]
# prints the graph-replayed rate and the per-kernel times of the eager pass for the shipped build and for every alternative, alternating
ALTS=$1; R=${2:-3}
for i in $(seq $R); do
  for lib in "" $ALTS; do
    KBN_LIB_PATH=$lib python bench.py --no-void --no-side-batch --no-fp32-mfma --no-fp16 --no-mixed --no-sustained --no-batch1 --no-options --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('[%s]' % (os.path.basename('$lib') or 'shipped'), d['value'], d['ms_per_step'], {k: v['us_per_step'] for k, v in sorted(pk.items()) if k in ('kb1_front','kb1_depth_front','conv_tail','s2d','conv_split','conv_split_upfold','conv_split_s2','conv_split_1x1s2')})"
  done
done
