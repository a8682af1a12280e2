#!/bin/bash
# round 5, visit w: the full line at the driver's flags with the new defaults (batches of 16, tail merged, oracle calls last, CPU baseline at the cgroup's size), twice
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05w}; out=gpurun_out/$tag; mkdir -p $out
for rep in 1; do
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_$rep.json 2> $out/bench_$rep.err; echo "bench rc=$?"
python - $out/bench_$rep.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); x = r.get("realistic") or {}; e = r.get("extra", {})
print("value", round(r["value"]), "ms/step", round(r["ms_per_step"], 4), "repeat", {k: round(v) for k, v in r["value_repeat_blocks"].items() if isinstance(v, float)}, "three streams", round(r["value_three_batch_streams"]["median"]), "per-step", round(r["value_one_moments_launch_per_step"]["median"]), "same pair", round(r["value_same_pair"]["median"]))
print("realistic", {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in x.items() if k in ("value", "value_with_attached_walk", "value_with_rounded_exact_means", "steps_per_block", "blocks", "error", "latency_ms_blocking", "rel_err_vs_oracle")})
print("roofline", r["roofline"]["bound"], round(r["roofline"]["frac"], 3), "alg", round(r["roofline"]["frac_algorithmic"], 3), "kernel_ms", round(r["roofline"]["kernel_ms"], 4), "sets", r["roofline"]["sets_per_launch"], "latency", r.get("latency_ms_blocking"), "parity", r.get("parity_rel_err_vs_cpu"))
print("cpu", {k: r["cpu_baseline"].get(k) for k in ("value", "cores", "scores_per_s_by_blas_threads")})
print("host_resident", e.get("host_resident", {}).get("scores_per_s"), "c4", e.get("c4_moments", {}).get("frac_of_8TBps"), "score_inf", e.get("score_inf_c3", {}).get("ms_batched_device_route"))
for k in ("per_song_config5_shape", "per_song_config5_encoder_frames", "per_song_config4_shape"): print(k, e.get(k, {}).get("ms"))
for k in ("k^-0.5", "k^-1", "k^-2"): print(k, {kk: e["frechet_decaying_c3"][k][kk] for kk in ("ms", "iterations", "rel_err_vs_oracle")})
PY
done
echo "== done"
