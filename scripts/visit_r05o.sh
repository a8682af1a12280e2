#!/bin/bash
# round 5, visit o: the bench line with the realistic block measured before the side blocks (driver flags), twice
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05o}; out=gpurun_out/$tag; mkdir -p $out
for rep in 1; do
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_$rep.json 2> $out/bench_$rep.err; echo "bench rc=$?"
python - $out/bench_$rep.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); x = r.get("realistic") or {}; e = r.get("extra", {})
print("value", round(r["value"]), "repeat", r["value_repeat_blocks"], "three streams", round(r["value_three_batch_streams"]["median"]), "per-step", round(r["value_one_moments_launch_per_step"]["median"]))
print("realistic", {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in x.items() if k in ("value", "value_with_attached_walk", "value_with_rounded_exact_means", "steps_per_block", "blocks", "error", "latency_ms_blocking", "rel_err_vs_oracle")})
print("roofline", r["roofline"]["bound"], round(r["roofline"]["frac"], 3), "kernel_ms", round(r["roofline"]["kernel_ms"], 4), "latency", r.get("latency_ms_blocking"), "host_resident", e.get("host_resident", {}).get("scores_per_s"), "c4", e.get("c4_moments", {}).get("frac_of_8TBps"))
for k in ("per_song_config5_shape", "per_song_config5_encoder_frames", "per_song_config4_shape"): print(k, e.get(k, {}).get("ms"))
PY
done
echo "== done"
