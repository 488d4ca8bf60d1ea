#!/bin/bash
# same-box A/B of one KBN_* switch inside the benchmark's forward: tools/ab_knob.sh KBN_NO_FUSED_SUB [repeats]
K=$1; R=${2:-3}
for i in $(seq $R); do
  for v in 0 1; do
    env $K=$v python bench.py --no-void --no-side-batch --no-bf16 --no-fp32-mfma --no-fp16 --no-mixed --no-sustained --no-batch1 --no-options --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('$K=$v', d['value'], d['ms_per_step'], {k: v['us_per_step'] for k, v in pk.items() if k.startswith(('conv_dma<1', 'kb1_front', 'kb_xyz'))})"
  done
done
