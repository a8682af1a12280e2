export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "numpys_mean or config3 or online or segment or runsum or running or score_inf or file_mean or shifted" > gpurun_out/r6i_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r6i_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r6i_bench.json 2> gpurun_out/r6i_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r6i_bench.json'))
print({k:d.get(k) for k in ('value','ms_per_step','parity_rel_err_vs_golden_g7','fad','value_realistic')})
print(d.get('value_repeat_blocks'))
print({k:(d.get('realistic') or {}).get(k) for k in ('parity_rel_err_vs_oracle','value_with_attached_walk','value_with_rounded_exact_means')})
PY
tail -3 gpurun_out/r6i_bench.err
