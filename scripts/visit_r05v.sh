#!/bin/bash
# round 5, visit v: pairs per batched chain 16 / 24 / 32 (steady state: 96 steps; and the driver's K = 20), then the full line with the default
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05v}; out=gpurun_out/$tag; mkdir -p $out
for b in 16 24 32; do
  timeout 400 python bench.py --realistic-only --batch $b --steps 96 --warmup 32 > $out/bench_b$b.json 2> $out/bench_b$b.err
  python - $out/bench_b$b.json $b <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); x = r.get("realistic") or {}
print("batch", sys.argv[2], "| value", round(r["value"]), "ms_per_step", round(r["ms_per_step"], 4), "| realistic", {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in x.items() if k in ("value", "value_with_attached_walk", "value_with_rounded_exact_means", "error", "rel_err_vs_oracle")}, "| fad", r.get("fad"))
PY
done 2>&1 | tee $out/batch_ab.txt
for b in 16 20 32; do
  timeout 400 python bench.py --timed-only --batch $b --steps 20 --warmup 5 > $out/bench_k20_b$b.json 2> $out/bench_k20_b$b.err
  python -c "
import json; r=json.load(open('$out/bench_k20_b$b.json')); print('K=20 batch $b value', round(r['value']), 'ms_per_step', round(r['ms_per_step'],4))" | tee -a $out/batch_ab.txt
done
timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "wide_chain or multi_job or jobs_in_flight or score_inf or tile256_kernel_matches" > $out/pytest_multi.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_multi.log | tail -5 | cut -c1-300
echo "== done"
