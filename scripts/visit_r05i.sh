#!/bin/bash
# round 5, the round's measured state: whole GPU suite, the bench line at the driver's flags, rocprofv3 kernel statistics + timeline of the TIMED
# loop at the same flags (bench.py --timed-only --steps 20 --warmup 5), separate FETCH_SIZE / WRITE_SIZE / SQ counter passes of the 8-matrix
# launches on rotated (HBM-streamed) inputs, the per-song calls by kernel and an SQ pass of the [1500 x 768] call, the realistic batch with three
# batches in flight, the probes, and clock / power samples (rocm-smi) under the tile kernel's MFMA-only ablation on Gaussian and on all-zero frames.
# rocprofv3 --kernel-trace [--pmc] only.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05i}
out=gpurun_out/$tag; mkdir -p $out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
rm -rf /tmp/prof_single
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_single -o bench -- python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$out/prof_single.json 2> $GRAFT_REPO_ROOT/$out/prof_single.err)
db=$(find /tmp/prof_single -name "*.db" | head -1)
if [ -n "$db" ]; then python scripts/rocpd_summary.py stats "$db" > $out/kernel_stats_single.csv; python scripts/rocpd_timeline.py "$db" 90 > $out/kernel_timeline_single.csv; fi
echo "== timed loop under rocprofv3"; grep "fad::" $out/kernel_stats_single.csv | cut -c1-130 | head -12
python -c "
import json; r=json.load(open('$out/prof_single.json')); print('   line under rocprof: value', round(r['value']), 'kernel_ms', round(r['roofline']['kernel_ms'],4), 'sets', r['roofline']['sets_per_launch'], 'frac', round(r['roofline']['frac'],3), r['roofline']['bound'])"
for c in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  name=$(echo $c | cut -d" " -f1)
  rm -rf /tmp/pmc_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$name -o b -- python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 12 --warmup 4 > /dev/null 2>&1); echo "pmc $name rc=$?"
  db=$(find /tmp/pmc_$name -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py pmc "$db" > $out/pmc_$name.csv
  grep -E "moments_tile256<0, false>|moments_reduce256" $out/pmc_$name.csv | head -16 | cut -c1-140
done
prof() {   # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- "$@" > $GRAFT_REPO_ROOT/$out/$name.txt 2>&1)
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/${name}_kernel_stats.csv
  echo "== $name"; grep "^mode" $out/$name.txt; grep "fad::" $out/${name}_kernel_stats.csv | cut -c1-120 | head -14
}
prof songs768 python $GRAFT_REPO_ROOT/scripts/probe_song_steps.py trace 768
prof songs128 python $GRAFT_REPO_ROOT/scripts/probe_song_steps.py trace 128
prof realistic_detached python $GRAFT_REPO_ROOT/scripts/probe_realistic.py detached pipelined
prof realistic_attached python $GRAFT_REPO_ROOT/scripts/probe_realistic.py attached pipelined
prof realistic_off python $GRAFT_REPO_ROOT/scripts/probe_realistic.py off pipelined
rm -rf /tmp/pmc_songs
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/pmc_songs -o b -- python $GRAFT_REPO_ROOT/scripts/probe_song_steps.py trace 768 > /dev/null 2>&1); echo "pmc songs rc=$?"
db=$(find /tmp/pmc_songs -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py pmc "$db" > $out/pmc_songs768_SQ.csv; grep "nsf_big\|nsf_i8_big" $out/pmc_songs768_SQ.csv | head -20 | cut -c1-140
timeout 300 python scripts/probe_runsum.py > $out/probe_runsum.txt 2>&1; grep "sets=\|differ\|two updates" $out/probe_runsum.txt
timeout 300 python scripts/probe_illcond.py > $out/probe_illcond.txt 2>&1; grep -E "spectrum|rror" $out/probe_illcond.txt | cut -c1-200
timeout 300 python scripts/probe_stall.py 300 > $out/probe_stall.txt 2>&1; grep -E "median|rror" $out/probe_stall.txt | cut -c1-300
timeout 300 python scripts/probe_guard.py > $out/probe_guard.txt 2>&1; grep "outlier" $out/probe_guard.txt | cut -c1-160
timeout 300 python scripts/probe_song_steps.py > $out/probe_song_steps.txt 2>&1; grep -v amdgpu $out/probe_song_steps.txt | cut -c1-220
# clocks and power under the MFMA-only ablation of the tile kernel
if timeout 300 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DT2_ABL_NODMA -DT2_ABL_NOREAD -o /tmp/t256_mfma scripts/probes/tile256_bench.hip 2> $out/t256_build.err; then
  T2_REPS=2500 bash scripts/smi_sample.sh "MFMA only (no LDS-DMA, no transpose reads), Gaussian frames, 2 x [400000 x 512]" /tmp/t256_mfma 512 400000 > $out/smi_mfma_gaussian.txt 2>&1; cat $out/smi_mfma_gaussian.txt | cut -c1-220
  T2_ZERO=1 T2_REPS=2500 bash scripts/smi_sample.sh "MFMA only, all-zero frames" /tmp/t256_mfma 512 400000 > $out/smi_mfma_zero.txt 2>&1; cat $out/smi_mfma_zero.txt | cut -c1-220
else echo "tile256_bench did not build"; tail -3 $out/t256_build.err; fi
if timeout 300 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/t256_full scripts/probes/tile256_bench.hip 2>> $out/t256_build.err; then
  T2_REPS=2500 bash scripts/smi_sample.sh "the full kernel, Gaussian frames" /tmp/t256_full 512 400000 > $out/smi_full_gaussian.txt 2>&1; cat $out/smi_full_gaussian.txt | cut -c1-220
fi
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_gpu.log | tail -8 | cut -c1-300
python - $out <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1] + "/bench.json"))
    print("value", round(r["value"]), "ms_per_step", round(r["ms_per_step"], 4), "repeat", round(r["value_repeat_blocks"]["median"]), "roofline", r["roofline"]["bound"], round(r["roofline"]["frac"], 3), "alg", round(r["roofline"]["frac_algorithmic"], 3), "issued", round(r["roofline"]["frac_issued"], 3), "kernel_ms", round(r["roofline"]["kernel_ms"], 4), "single", r["roofline"]["single_score_launch"], "parity", r.get("parity_rel_err_vs_cpu"), "latency", r.get("latency_ms_blocking"), "cpu", r.get("cpu_baseline", {}).get("value"))
    x = r.get("realistic") or {}
    print("realistic", {k: x.get(k) for k in x if k != "workload"})
    e = r.get("extra", {})
    print("c4", {k: e.get("c4_moments", {}).get(k) for k in ("ms", "frac_of_8TBps", "tile_kernel_frac_of_8TBps", "one_update_of_all_files", "with_reference_order_file_means")})
    for k in ("per_song_config5_shape", "per_song_config5_encoder_frames", "per_song_config4_shape"): print(k, {kk: e.get(k, {}).get(kk) for kk in ("ms", "songs_per_s", "ok", "ms_spread")})
    for k in ("k^-0.5", "k^-1", "k^-2"): print(k, {kk: e["frechet_decaying_c3"][k][kk] for kk in ("ms", "iterations", "route", "rel_err_vs_oracle", "ms_spread")})
    print("host_resident", e.get("host_resident", {}).get("scores_per_s"), "score_inf", e.get("score_inf_c3", {}).get("ms_batched_device_route"), e.get("score_inf_c3", {}).get("max_rel_err_vs_oracle_sample"))
except Exception as ex:
    print("bench line unreadable:", ex)
PY
echo "== done"
