#!/bin/bash
# same-box A/B of one bench.py command-line flag inside the benchmark's forward: tools/ab_flag.sh --split-graphs [repeats]
F=$1; R=${2:-3}
for i in $(seq $R); do
  for v in "" "$F"; do
    python bench.py $v --no-void --no-side-batch --no-bf16 --no-fp32-mfma --no-fp16 --no-mixed --no-sustained --no-batch1 --no-options --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'], d['config']['launch'])"
  done
done
