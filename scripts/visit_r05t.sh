#!/bin/bash
# round 5, visit t: the rebuilt library (walk kernel changes, 16-pair batches) -- the walk alone, the whole GPU suite, batch 8 vs 16, the stall probe with scheduler statistics
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05t}; out=gpurun_out/$tag; mkdir -p $out
strings fadtk_amd/lib/libfad_hip.so | grep "count=%d out of range"
timeout 300 python scripts/probe_runsum.py > $out/probe_runsum.txt 2>&1; grep "sets=\|differ\|two updates" $out/probe_runsum.txt
for b in 8 16; do
  timeout 400 python bench.py --realistic-only --batch $b --steps 64 --warmup 16 > $out/bench_b$b.json 2> $out/bench_b$b.err
  python - $out/bench_b$b.json $b <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); x = r.get("realistic") or {}
print("batch", sys.argv[2], "| value", round(r["value"]), "ms_per_step", round(r["ms_per_step"], 4), "| realistic", {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in x.items() if k in ("value", "value_with_attached_walk", "value_with_rounded_exact_means", "error", "rel_err_vs_oracle")}, "| fad", r.get("fad"))
PY
done 2>&1 | tee $out/batch_ab.txt
for b in 8 16; do
  timeout 400 python bench.py --timed-only --batch $b --steps 20 --warmup 5 > $out/bench_k20_b$b.json 2> $out/bench_k20_b$b.err
  python -c "
import json; r=json.load(open('$out/bench_k20_b$b.json')); print('K=20 batch $b value', round(r['value']), 'ms_per_step', round(r['ms_per_step'],4))" | tee -a $out/batch_ab.txt
done
timeout 300 python scripts/probe_stall.py 300 2>&1 | grep -E "median|collector|cgroup" | cut -c1-400 | tee $out/probe_stall.txt
OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1 timeout 300 python scripts/probe_stall.py 300 2>&1 | grep -E "median|collector|cgroup" | cut -c1-400 | sed 's/^/1 BLAS thread: /' | tee -a $out/probe_stall.txt
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_gpu.log | tail -8 | cut -c1-300
echo "== done"
