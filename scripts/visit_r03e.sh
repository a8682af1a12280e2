#!/bin/bash
# round 3, visit e: full suite + full bench line (extras, CPU baseline) with the eight-launch chain
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03e; mkdir -p $out
echo "== full gpu test suite"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $out/pytest_gpu.log
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
o = json.load(open("gpurun_out/r03e/bench.json"))
ex = o.pop("extra", {})
print(json.dumps({k: o[k] for k in ("value", "ms_per_step", "breakdown_ms", "value_repeat_blocks", "value_same_pair", "value_stream_per_score", "fad", "parity_rel_err_vs_cpu", "newton_schulz_iters", "speedup_vs_cpu")}, indent=0))
print("roofline", {k: o["roofline"][k] for k in ("frac", "kernel_ms", "mfma_util")})
print("roofline_frechet", {k: o["roofline_frechet"][k] for k in ("ms", "frac", "gemms")})
for k, v in ex.items():
    print(k, json.dumps({kk: vv for kk, vv in v.items() if kk not in ("roofline", "note", "cpu_baseline")})[:700])
PY
tail -3 $out/bench.err
echo "== done"
