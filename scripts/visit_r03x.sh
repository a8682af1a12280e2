#!/bin/bash
# round 3, final state: whole GPU suite, the bench line, rocprofv3 kernel statistics + launch sequence of the bench command in both
# stream layouts, separate FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --kernel-trace --pmc only)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r03x}
out=gpurun_out/$tag; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest_gpu.log | cut -c1-300
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - $out <<'PY'
import json, sys
r = json.load(open(sys.argv[1] + "/bench.json"))
print("value", round(r["value"]), "ms_per_step", round(r["ms_per_step"], 4), "repeat", round(r["value_repeat_blocks"]["median"]), "roofline", round(r["roofline"]["achieved"]), round(r["roofline"]["frac"], 3),
      "alone", r["roofline"].get("alone"), "frechet", r["breakdown_ms"], "parity", r.get("parity_rel_err_vs_cpu"), "cpu", r.get("cpu_baseline", {}).get("value"))
for k, v in r.get("extra", {}).items():
    if isinstance(v, dict):
        print(" ", k, {kk: vv for kk, vv in v.items() if kk in ("ms", "songs_per_s", "ok", "max_rel_err_vs_oracle_sample", "scores_per_s", "ms_batched_device_route", "frac_of_8TBps", "one_update_of_all_files", "error")})
PY
for mode in lanes single; do
  flag=""; [ $mode = single ] && flag="--single-stream"
  rm -rf /tmp/prof_$mode
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $flag > /dev/null 2> $GRAFT_REPO_ROOT/$out/prof_$mode.err)
  db=$(find /tmp/prof_$mode -name "*.db" | head -1)
  if [ -n "$db" ]; then python scripts/rocpd_summary.py stats "$db" > $out/kernel_stats_$mode.csv; python scripts/rocpd_summary.py seq "$db" 60 > $out/kernel_sequence_$mode.csv; fi
  echo "== $mode"; grep "fad::" $out/kernel_stats_$mode.csv | cut -c1-120
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --single-stream > /dev/null 2>&1); echo "pmc $c rc=$?"
  db=$(find /tmp/pmc_$c -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py pmc "$db" > $out/pmc_$c.csv
  grep -E "moments_tile|vectorized_elementwise|nsf_" $out/pmc_$c.csv | head -12 | cut -c1-140
done
echo "== done"
