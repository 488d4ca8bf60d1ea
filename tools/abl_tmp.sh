for lib in "" alt/nowl.so alt/nomfma.so alt/noboth.so; do KBN_LIB_PATH=$lib python bench.py --no-void --no-side-batch --no-fp32-mfma --no-fp16 --no-mixed --no-sustained --no-batch1 --no-options --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('[%s]' % (os.path.basename('$lib') or 'shipped'), d['value'], pk['deconv0_tail']['us_per_step'])"; done
