#!/bin/bash
# round 5, seventh visit: the walk back on 16 columns per workgroup with the guard's second pass held to 224 registers -- the walk alone, the
# realistic batch (detached / attached / off) under rocprofv3 --kernel-trace --stats, what the guard costs when it fires, the tests that touch
# the moments, the bench line.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05g}
out=gpurun_out/$tag; mkdir -p $out
timeout 300 python scripts/probe_runsum.py > $out/probe_runsum.txt 2>&1; grep "sets=\|differ\|two updates" $out/probe_runsum.txt
for mode in detached attached off; do
  rm -rf /tmp/prof_x
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $GRAFT_REPO_ROOT/scripts/probe_realistic.py $mode > $GRAFT_REPO_ROOT/$out/realistic_$mode.txt 2>&1)
  db=$(find /tmp/prof_x -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/realistic_${mode}_kernel_stats.csv
  grep "^mode" $out/realistic_$mode.txt; grep "colsum\|tile256\|reduce256" $out/realistic_${mode}_kernel_stats.csv | cut -c1-110
done
timeout 300 python scripts/probe_guard.py > $out/probe_guard.txt 2>&1; grep "outlier" $out/probe_guard.txt | cut -c1-160
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "moments or tile256 or guard or numpys or shifted or embd or statistics or online or songs or individual" > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_gpu.log | tail -12 | cut -c1-300
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - $out <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1] + "/bench.json"))
    print("value", round(r["value"]), "ms_per_step", round(r["ms_per_step"], 4), "roofline", r["roofline"]["bound"], round(r["roofline"]["frac"], 3), "parity", r.get("parity_rel_err_vs_cpu"), "latency", r.get("latency_ms_blocking"))
    x = r.get("realistic") or {}
    print("realistic", {k: x.get(k) for k in ("value", "value_with_attached_walk", "value_with_rounded_exact_means", "reference_order_mean_cost", "reference_order_mean_cost_attached", "latency_ms_blocking", "route", "iterations", "rel_err_vs_oracle", "rel_err_vs_oracle_with_rounded_exact_means", "error")})
    e = r.get("extra", {})
    print("c4", {k: e.get("c4_moments", {}).get(k) for k in ("ms", "frac_of_8TBps", "with_reference_order_file_means")})
    for k in ("per_song_config5_shape", "per_song_config5_encoder_frames", "per_song_config4_shape"): print(k, e.get(k, {}).get("ms"))
    for k in ("k^-0.5", "k^-1", "k^-2"): print(k, {kk: e["frechet_decaying_c3"][k][kk] for kk in ("ms", "iterations", "route", "rel_err_vs_oracle")})
    print("host_resident", e.get("host_resident", {}).get("scores_per_s"), "score_inf", e.get("score_inf_c3", {}).get("ms_batched_device_route"))
except Exception as ex:
    print("bench line unreadable:", ex)
PY
echo "== done"
