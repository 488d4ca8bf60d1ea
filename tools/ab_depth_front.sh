for i in 1 2 3; do
  for f in 0 1; do
    KBN_DEPTH_FRONT_FUSION=$f python bench.py --no-void --no-side-batch --no-bf16 --no-fp32-mfma --no-fp16 --no-mixed --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fusion=$f', d['value'], d['ms_per_step'])"
  done
done
