export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
for mg in 4 8 16; do
timeout 600 python bench.py --steps 20 --warmup 5 --realistic-only --no-cpu-baseline --moments-group $mg > gpurun_out/r6j_bench_$mg.json 2> gpurun_out/r6j_bench.err; echo "mg=$mg rc=$?"; python - $mg <<'PY'
import json,sys
d=json.load(open('gpurun_out/r6j_bench_%s.json'%sys.argv[1]))
r=d.get('realistic') or {}
print({k:d.get(k) for k in ('value','value_realistic')}, {k:r.get(k) for k in ('value_with_attached_walk','value_with_rounded_exact_means','parity_rel_err_vs_oracle','walk_ms','tile_kernel_ms')})
PY
done
tail -3 gpurun_out/r6j_bench.err
