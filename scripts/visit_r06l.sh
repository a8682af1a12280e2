export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -q -x --tb=short -p no:cacheprovider -k "prepared or dist or bench" > gpurun_out/r6l_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r6l_pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r6l_bench.json 2> gpurun_out/r6l_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r6l_bench.json'))
print({k:d.get(k) for k in ('value','ms_per_step','parity_rel_err_vs_golden_g7','value_realistic')}, d.get('value_repeat_blocks',{}).get('median'), (d.get('roofline') or {}).get('frac'))
print(d.get('host_ms_per_step'))
PY
done
tail -3 gpurun_out/r6l_bench.err
