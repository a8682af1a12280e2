export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "moments or config3 or statistics or tile256 or guard" > gpurun_out/r6e_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r6e_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6e_bench.json 2> gpurun_out/r6e_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r6e_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','value_repeat_blocks') if k in d})
print(json.dumps(d.get('roofline'),indent=None)[:1200])
print(json.dumps(d.get('roofline_frechet'),indent=None)[:600])
PY
tail -3 gpurun_out/r6e_bench.err
