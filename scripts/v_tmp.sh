for e in "--moments-stream" "" "--moments-stream" "--single-stream"; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras $e 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('[$e] value',round(j['value']),'median',round(j['value_repeat_median']),'kernel_ms',round(r['kernel_ms'],4),'frac',round(r['frac'],3),'issued',round(r['mfma_util'],3),'fad',j['fad_last_timed_step'])
"; done
