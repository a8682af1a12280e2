#!/bin/bash
# round 5, last visit: build check as the driver does it, smoke(), the whole GPU suite, the line at the driver's flags
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r06b}; out=gpurun_out/$tag; mkdir -p $out
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $out/smoke.txt
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_gpu.log | tail -8 | cut -c1-300
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - $out/bench.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); x = r.get("realistic") or {}; e = r.get("extra", {})
print("value", round(r["value"]), "ms/step", round(r["ms_per_step"], 4), "repeat", {k: round(v) for k, v in r["value_repeat_blocks"].items() if isinstance(v, float)})
print("realistic", {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in x.items() if k in ("value", "value_with_attached_walk", "value_with_rounded_exact_means", "blocks", "error", "latency_ms_blocking", "rel_err_vs_oracle")})
print("roofline", r["roofline"]["bound"], round(r["roofline"]["frac"], 3), "alg", round(r["roofline"]["frac_algorithmic"], 3), "kernel_ms", round(r["roofline"]["kernel_ms"], 4), "latency", r.get("latency_ms_blocking"), "parity", r.get("parity_rel_err_vs_cpu"), "cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
print("host_resident", e.get("host_resident", {}).get("scores_per_s"), "c4", e.get("c4_moments", {}).get("frac_of_8TBps"), "score_inf", e.get("score_inf_c3", {}).get("ms_batched_device_route"))
for k in ("per_song_config5_shape", "per_song_config5_encoder_frames", "per_song_config4_shape"): print(k, e.get(k, {}).get("ms"))
for k in ("k^-0.5", "k^-1", "k^-2"): print(k, {kk: e["frechet_decaying_c3"][k].get(kk) for kk in ("ms", "iterations", "rel_err_vs_oracle", "ms_per_score_in_a_batch_of_16", "batch_route", "batch_rel_err_vs_oracle")})
PY
echo "== done"
