#!/bin/bash
# same-box A/B of one KBN_DEBUG bit inside the benchmark's forward: tools/ab_debug_bit.sh 1024 [repeats]   (1024: deconv0 as two launches)
B=$1; R=${2:-3}
for i in $(seq $R); do
  for v in 0 $B; do
    KBN_DEBUG=$v python bench.py --no-void --no-side-batch --no-fp32-mfma --no-fp16 --no-mixed --no-sustained --no-batch1 --no-options --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('KBN_DEBUG=$v', d['value'], d['ms_per_step'], {k: v['us_per_step'] for k, v in pk.items() if k.startswith(('conv_tail', 'deconv0', 'conv_split_upfold'))})"
  done
done
