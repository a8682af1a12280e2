#!/bin/bash
# round 5, visit s: batches of 16 pairs per chain (A/B against 8 in the bench's flat and realistic loops) + the multi-pair tests; the stall probe under --hip-trace, twice
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05s}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "wide_chain or multi_job or jobs_in_flight or score_inf" > $out/pytest_multi.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_multi.log | tail -6 | cut -c1-300
for b in 8 16; do
  timeout 400 python bench.py --realistic-only --batch $b --steps 64 --warmup 16 > $out/bench_b$b.json 2> $out/bench_b$b.err
  python - $out/bench_b$b.json $b <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); x = r.get("realistic") or {}
print("batch", sys.argv[2], "| value", round(r["value"]), "ms_per_step", round(r["ms_per_step"], 4), "| realistic", {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in x.items() if k in ("value", "value_with_attached_walk", "value_with_rounded_exact_means", "error", "rel_err_vs_oracle")}, "| parity fad", r.get("fad"))
PY
done 2>&1 | tee $out/batch_ab.txt
for rep in 1 2; do
  rm -rf /tmp/prof_stall
  (cd /tmp && timeout 300 rocprofv3 --hip-trace --kernel-trace -d /tmp/prof_stall -o s -- python $GRAFT_REPO_ROOT/scripts/probe_stall.py 300 > $GRAFT_REPO_ROOT/$out/probe_stall_$rep.txt 2>&1)
  grep -E "median|collector" $out/probe_stall_$rep.txt | cut -c1-300
  db=$(find /tmp/prof_stall -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_long_calls.py "$db" 5 > $out/stall_long_calls_$rep.txt 2>&1; grep -A4 "^== regions " $out/stall_long_calls_$rep.txt | cut -c1-200
done
echo "== done"
