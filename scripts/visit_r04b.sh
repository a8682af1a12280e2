#!/bin/bash
# round 4: full GPU suite + bench line + probe on the ping-pong 256-column-slab kernel
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r04b}
out=gpurun_out/$tag; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_gpu.log | tail -8 | cut -c1-300
timeout 300 python scripts/probe_tile256.py > $out/probe_tile256.txt 2>&1; grep "x" $out/probe_tile256.txt | cut -c1-200
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - $out <<'PY'
import json, sys
r = json.load(open(sys.argv[1] + "/bench.json"))
print("value", round(r["value"]), "ms_per_step", round(r["ms_per_step"], 4), "repeat", round(r["value_repeat_blocks"]["median"]), "roofline", r["roofline"]["kernel"], round(r["roofline"]["kernel_ms"], 4), round(r["roofline"]["frac"], 3),
      "alone", r["roofline"].get("alone"), "breakdown", r["breakdown_ms"], "parity", r.get("parity_rel_err_vs_cpu"), "cpu", r.get("cpu_baseline", {}).get("value"))
print("single", r.get("value_single_stream"), "same_pair", r.get("value_same_pair"))
for k, v in r.get("extra", {}).items():
    if isinstance(v, dict):
        print(" ", k, {kk: vv for kk, vv in v.items() if kk in ("ms", "songs_per_s", "ok", "max_rel_err_vs_oracle_sample", "scores_per_s", "ms_batched_device_route", "frac_of_8TBps", "one_update_of_all_files", "error")})
PY
echo "== done"
