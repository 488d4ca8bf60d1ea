#!/bin/bash
# graph branch count inside the benchmark's forward: tools/ab_branches.sh "1 2 4" [repeats]
for i in $(seq ${2:-2}); do
  for b in $1; do
    python bench.py --branches $b --no-void --no-side-batch --no-fp32-mfma --no-fp16 --no-mixed --no-sustained --no-batch1 --no-options --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('branches $b', d['value'], d['ms_per_step'])"
  done
done
