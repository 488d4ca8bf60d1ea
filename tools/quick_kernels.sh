#!/bin/bash
# per-kernel-group times of the benchmark's forward (eager pass, HIP events) + the graph-replayed rate, N passes: tools/quick_kernels.sh [passes]
for i in $(seq ${1:-2}); do
python bench.py --no-void --no-side-batch --no-bf16 --no-fp32-mfma --no-fp16 --no-mixed --no-sustained --no-batch1 --no-options --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print(d['value'], d['ms_per_step'], {k: v['us_per_step'] for k, v in sorted(pk.items())})"
done
