#!/bin/bash
# round 4, final: the batched schedule on ONE stream is the default layout (bench.py; --multi-stream = three batch streams): rocprofv3
# kernel statistics + timeline of the TIMED loop only (bench.py --timed-only) in the shipped schedule (three batch streams) and with
# everything on one stream, separate FETCH_SIZE / WRITE_SIZE and SQ counter passes of the 8-matrix launches on rotated (HBM-streamed)
# inputs, the probes, the bench line, [the whole GPU suite].  rocprofv3 --kernel-trace [--pmc] only.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r04i}
out=gpurun_out/$tag; mkdir -p $out
if [ "$2" != "noprof" ]; then
for mode in single multi; do
  flag=""; [ $mode = multi ] && flag="--multi-stream"
  rm -rf /tmp/prof_$mode
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o bench -- python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 48 --warmup 8 $flag > $GRAFT_REPO_ROOT/$out/prof_$mode.json 2> $GRAFT_REPO_ROOT/$out/prof_$mode.err)
  db=$(find /tmp/prof_$mode -name "*.db" | head -1)
  if [ -n "$db" ]; then python scripts/rocpd_summary.py stats "$db" > $out/kernel_stats_$mode.csv; python scripts/rocpd_timeline.py "$db" 90 > $out/kernel_timeline_$mode.csv; fi
  echo "== $mode"; grep "fad::" $out/kernel_stats_$mode.csv | cut -c1-130 | head -12
  python -c "
import json; r=json.load(open('$out/prof_$mode.json')); print('   line under rocprof: value', round(r['value']), 'kernel_ms', round(r['roofline']['kernel_ms'],4), 'sets', r['roofline']['sets_per_launch'], 'frac', round(r['roofline']['frac'],3))"
done
for c in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  name=$(echo $c | cut -d" " -f1)
  rm -rf /tmp/pmc_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$name -o b -- python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 12 --warmup 4 --single-stream > /dev/null 2>&1); echo "pmc $name rc=$?"
  db=$(find /tmp/pmc_$name -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py pmc "$db" > $out/pmc_$name.csv
  grep -E "moments_tile256<0, false>|moments_reduce256" $out/pmc_$name.csv | head -16 | cut -c1-140
done
timeout 300 python scripts/probe_sets.py > $out/probe_sets.txt 2>&1; grep "sets per" $out/probe_sets.txt | cut -c1-200
timeout 300 python scripts/probe_illcond.py > $out/probe_illcond.txt 2>&1; grep "spectrum" $out/probe_illcond.txt | cut -c1-200
fi
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - $out <<'PY'
import json, sys
r = json.load(open(sys.argv[1] + "/bench.json"))
print("value", round(r["value"]), "ms_per_step", round(r["ms_per_step"], 4), "repeat", round(r["value_repeat_blocks"]["median"]), "roofline", r["roofline"]["kernel"], round(r["roofline"]["kernel_ms"], 4), r["roofline"]["sets_per_launch"], round(r["roofline"]["frac"], 3),
      "traffic", r["roofline"]["traffic"], "alone", r["roofline"].get("alone"), "breakdown", r["breakdown_ms"], "parity", r.get("parity_rel_err_vs_cpu"), "cpu", r.get("cpu_baseline", {}).get("value"))
print("three batch streams", r.get("value_three_batch_streams"), "one launch per step", r.get("value_one_moments_launch_per_step"), "same_pair", r.get("value_same_pair"))
for k, v in r.get("extra", {}).items():
    if isinstance(v, dict):
        print(" ", k, {kk: vv for kk, vv in v.items() if kk in ("ms", "songs_per_s", "ok", "max_rel_err_vs_oracle_sample", "scores_per_s", "ms_batched_device_route", "frac_of_8TBps", "one_update_of_all_files", "error", "k^-0.5", "k^-1", "k^-2")})
PY
if [ "$3" = "tests" ]; then
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_gpu.log | tail -8 | cut -c1-300
fi
echo "== done"
