#!/bin/bash
# round 5, the round's final measured state at the defaults the line now uses (batches of 16 pairs per chain): the bench line at the driver's
# flags, rocprofv3 kernel statistics + timeline of the TIMED loop at the same flags, the realistic batch by kernel in its three modes, the probes
# that changed since r05i, and the whole GPU suite.  rocprofv3 --kernel-trace only (the counter passes of r05i still describe the 8-matrix launch).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05z}
out=gpurun_out/$tag; mkdir -p $out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
rm -rf /tmp/prof_single
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_single -o bench -- python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$out/prof_single.json 2> $GRAFT_REPO_ROOT/$out/prof_single.err)
db=$(find /tmp/prof_single -name "*.db" | head -1)
if [ -n "$db" ]; then python scripts/rocpd_summary.py stats "$db" > $out/kernel_stats_single.csv; python scripts/rocpd_timeline.py "$db" 120 > $out/kernel_timeline_single.csv; fi
echo "== timed loop under rocprofv3"; grep "fad::" $out/kernel_stats_single.csv | cut -c1-130 | head -12
python -c "
import json; r=json.load(open('$out/prof_single.json')); print('   line under rocprof: value', round(r['value']), 'kernel_ms', round(r['roofline']['kernel_ms'],4), 'sets', r['roofline']['sets_per_launch'], 'frac', round(r['roofline']['frac'],3), r['roofline']['bound'])"
prof() {   # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- "$@" > $GRAFT_REPO_ROOT/$out/$name.txt 2>&1)
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/${name}_kernel_stats.csv
  echo "== $name"; grep "^extra" $out/$name.txt | cut -c1-130; grep "fad::" $out/${name}_kernel_stats.csv | cut -c1-120 | head -8
}
prof realistic_detached python $GRAFT_REPO_ROOT/scripts/probe_realistic.py detached pipelined
prof realistic_attached python $GRAFT_REPO_ROOT/scripts/probe_realistic.py attached pipelined
prof realistic_off python $GRAFT_REPO_ROOT/scripts/probe_realistic.py off pipelined
timeout 300 python scripts/probe_runsum.py > $out/probe_runsum.txt 2>&1; grep "sets=\|differ\|two updates" $out/probe_runsum.txt
timeout 300 python scripts/probe_illcond.py > $out/probe_illcond.txt 2>&1; grep -E "spectrum|rror" $out/probe_illcond.txt | head -7 | cut -c1-200
timeout 200 python scripts/probe_host_pieces.py 2>&1 | grep "PIECE_KB\|update" | tee $out/probe_host_pieces.txt
timeout 300 python scripts/probe_stall.py 300 2>&1 | grep -E "median|collector|cgroup" | cut -c1-400 | tee $out/probe_stall.txt
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_gpu.log | tail -16 | cut -c1-300
python - $out <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1] + "/bench.json"))
    print("value", round(r["value"]), "ms_per_step", round(r["ms_per_step"], 4), "repeat", {k: round(v) for k, v in r["value_repeat_blocks"].items() if isinstance(v, float)}, "roofline", r["roofline"]["bound"], round(r["roofline"]["frac"], 3), "alg", round(r["roofline"]["frac_algorithmic"], 3), "issued", round(r["roofline"]["frac_issued"], 3), "kernel_ms", round(r["roofline"]["kernel_ms"], 4), "parity", r.get("parity_rel_err_vs_cpu"), "latency", r.get("latency_ms_blocking"), "cpu", r.get("cpu_baseline", {}).get("value"), r.get("cpu_baseline", {}).get("cores"), r.get("cpu_baseline", {}).get("scores_per_s_by_blas_threads"))
    x = r.get("realistic") or {}
    print("realistic", {k: x.get(k) for k in x if k != "workload"})
    e = r.get("extra", {})
    print("c4", {k: e.get("c4_moments", {}).get(k) for k in ("ms", "frac_of_8TBps", "tile_kernel_frac_of_8TBps")}, "with ref file means", e.get("c4_moments", {}).get("with_reference_order_file_means", {}).get("ms"))
    for k in ("per_song_config5_shape", "per_song_config5_encoder_frames", "per_song_config4_shape"): print(k, {kk: e.get(k, {}).get(kk) for kk in ("ms", "songs_per_s", "ok")})
    for k in ("k^-0.5", "k^-1", "k^-2"): print(k, {kk: e["frechet_decaying_c3"][k][kk] for kk in ("ms", "iterations", "route", "rel_err_vs_oracle")})
    print("host_resident", e.get("host_resident", {}).get("scores_per_s"), "score_inf", e.get("score_inf_c3", {}).get("ms_batched_device_route"), e.get("score_inf_c3", {}).get("max_rel_err_vs_oracle_sample"))
except Exception as ex:
    print("bench line unreadable:", ex)
PY
echo "== done"
