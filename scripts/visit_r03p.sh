#!/bin/bash
# round 3, visit p: whole GPU suite + the full bench line after the per-song work
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03p; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest_gpu.log | cut -c1-300
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r03p/bench.json"))
print("value", r["value"], "ms_per_step", r["ms_per_step"], "roofline", r["roofline"]["achieved"], r["roofline"]["frac"])
e = r.get("extra", {})
for k, v in e.items():
    if isinstance(v, dict):
        print(k, {kk: vv for kk, vv in v.items() if kk in ("ms", "songs_per_s", "ok", "max_rel_err_vs_oracle_sample", "scores_per_s", "median", "value", "GBps", "GBps_frames", "TBps")})
PY
echo "== done"
