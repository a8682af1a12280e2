#!/bin/bash
# round 5, fifth visit (walk on half the chip, segment walks on the LDS-staged kernel, code objects warmed, songs on the lean correction kernel): kernel statistics (rocprofv3 --kernel-trace --stats) of the per-song calls and of the realistic batch, the walk
# probe, decaying pairs, the stall hunt (rocprofv3 --hip-trace --kernel-trace around scripts/probe_stall.py), selected GPU tests, the bench line.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05e}
out=gpurun_out/$tag; mkdir -p $out
prof() {   # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- "$@" > $GRAFT_REPO_ROOT/$out/$name.txt 2>&1)
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/${name}_kernel_stats.csv
  echo "== $name"; grep -v "amdgpu.ids" $out/$name.txt | tail -2 | cut -c1-250; grep "fad::" $out/${name}_kernel_stats.csv | cut -c1-120 | head -16
}
prof songs768 python $GRAFT_REPO_ROOT/scripts/probe_song_steps.py trace 768
prof songs128 python $GRAFT_REPO_ROOT/scripts/probe_song_steps.py trace 128
prof realistic_detached python $GRAFT_REPO_ROOT/scripts/probe_realistic.py detached
prof realistic_attached python $GRAFT_REPO_ROOT/scripts/probe_realistic.py attached
prof realistic_off python $GRAFT_REPO_ROOT/scripts/probe_realistic.py off
timeout 300 python scripts/probe_runsum.py > $out/probe_runsum_side.txt 2>&1; grep "sets=\|differ\|two updates" $out/probe_runsum_side.txt
timeout 300 python scripts/probe_illcond.py > $out/probe_illcond.txt 2>&1; grep -E "spectrum|rror" $out/probe_illcond.txt | cut -c1-200
# the stall hunt
rm -rf /tmp/prof_stall
(cd /tmp && timeout 600 rocprofv3 --hip-trace --kernel-trace -d /tmp/prof_stall -o s -- python $GRAFT_REPO_ROOT/scripts/probe_stall.py 300 > $GRAFT_REPO_ROOT/$out/probe_stall.txt 2>&1); echo "stall probe rc=$?"
grep -E "median|rror" $out/probe_stall.txt | cut -c1-400
db=$(find /tmp/prof_stall -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_long_calls.py "$db" 3 > $out/stall_long_calls.txt 2>&1; head -60 $out/stall_long_calls.txt | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_gpu.log | tail -25 | cut -c1-300
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
tail -3 $out/bench.err | cut -c1-300
echo "== done"
