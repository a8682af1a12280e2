export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "frechet or thread or golden or g10 or score or song or config" 2>&1 | tail -3
timeout 600 python scripts/probe_illcond.py 2>&1 | grep "D="
timeout 300 python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('value',round(d['value'],1),'ms/step',round(d['ms_per_step'],4),'kernel_ms',round(d['roofline']['kernel_ms'],5),'frac',round(d['roofline']['frac'],4),'fad',d['fad'])"
