export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
for mg in 4 8 16 4 8 16; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --moments-group $mg > gpurun_out/r6f_bench_$mg.json 2> gpurun_out/r6f_bench.err; echo "mg=$mg rc=$?"; python - $mg <<'PY'
import json,sys
d=json.load(open('gpurun_out/r6f_bench_%s.json'%sys.argv[1]))
r=d.get('roofline',{})
print({k:d[k] for k in ('value','ms_per_step') if k in d}, d.get('value_repeat_blocks',{}).get('median'), 'frac',r.get('frac'), 'GB/s', r.get('achieved'), 'parity', d.get('parity_rel_err_vs_oracle', d.get('parity')))
PY
done
tail -3 gpurun_out/r6f_bench.err
