export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for L in 1 2 3 4; do
  timeout 300 python bench.py --steps 100 --warmup 8 --inflight $L --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('inflight',d['scores_in_flight'],'value',round(d['value'],1),'ms/step',round(d['ms_per_step'],4),'kernel_ms',round(d['roofline']['kernel_ms'],5),'frac',round(d['roofline']['frac'],4),'fad',d['fad'],d['step_ms_spread'])"
done
FAD_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('dist inflight',d['scores_in_flight'],'value',round(d['value'],1),'ms/step',round(d['ms_per_step'],4),'kernel_ms',round(d['roofline']['kernel_ms'],5),'frac',round(d['roofline']['frac'],4),'fad',d['fad'])"
timeout 300 python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('steps 3 warmup 0: value',round(d['value'],1),'fad',d['fad'])"
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('steps 1: value',round(d['value'],1),'fad',d['fad'])"
