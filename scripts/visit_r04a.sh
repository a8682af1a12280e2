#!/bin/bash
# round 4, first visit: the 256-column-slab moments kernel -- parity tests, A/B against the 128 x 128 kernel, bench line (timed
# loop only) with both kernels, rocprofv3 kernel statistics of the timed loop.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r04a}
out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "moments" > $out/pytest_moments.log 2>&1; echo "pytest moments rc=$?"; tail -15 $out/pytest_moments.log | cut -c1-400
timeout 600 python scripts/probe_tile256.py > $out/probe_tile256.txt 2>&1; echo "probe rc=$?"; cat $out/probe_tile256.txt | cut -c1-220
for knob in 1 0; do
  FAD_MOMENTS_TILE256=$knob timeout 300 python bench.py --timed-only --steps 30 --warmup 5 > $out/bench_timed_t256_$knob.json 2> $out/bench_timed_$knob.err; echo "bench timed-only (tile256=$knob) rc=$?"
  python - $out/bench_timed_t256_$knob.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("  value", round(r["value"]), "ms_per_step", round(r["ms_per_step"], 4), "kernel", r["roofline"]["kernel"], "kernel_ms", round(r["roofline"]["kernel_ms"], 4), "frac", round(r["roofline"]["frac"], 3), "mfma_util", round(r["roofline"]["mfma_util"], 3), "fad", r["fad"])
PY
done
timeout 300 python bench.py --timed-only --single-stream --steps 30 --warmup 5 > $out/bench_timed_single.json 2>/dev/null; python -c "
import json; r=json.load(open('$out/bench_timed_single.json')); print('single stream: value', round(r['value']), 'ms', round(r['ms_per_step'],4), 'kernel_ms', round(r['roofline']['kernel_ms'],4), 'frac', round(r['roofline']['frac'],3))"
for cc in 4 8; do
  timeout 300 python bench.py --timed-only --chain-cus $cc --steps 30 --warmup 5 > $out/bench_timed_cc$cc.json 2> $out/bench_cc$cc.err; echo "chain-cus $cc rc=$?"
  python -c "
import json; r=json.load(open('$out/bench_timed_cc$cc.json')); print('chain-cus $cc: value', round(r['value']), 'ms', round(r['ms_per_step'],4), 'kernel_ms', round(r['roofline']['kernel_ms'],4), 'frac', round(r['roofline']['frac'],3))" 2>/dev/null || tail -3 $out/bench_cc$cc.err
done
for mode in lanes single; do
  flag=""; [ $mode = single ] && flag="--single-stream"
  rm -rf /tmp/prof_$mode
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o bench -- python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 30 --warmup 5 $flag > $GRAFT_REPO_ROOT/$out/prof_$mode.json 2> $GRAFT_REPO_ROOT/$out/prof_$mode.err)
  db=$(find /tmp/prof_$mode -name "*.db" | head -1)
  if [ -n "$db" ]; then python scripts/rocpd_summary.py stats "$db" > $out/kernel_stats_$mode.csv; python scripts/rocpd_summary.py seq "$db" 60 > $out/kernel_sequence_$mode.csv; fi
  echo "== $mode"; grep "fad::" $out/kernel_stats_$mode.csv | cut -c1-130
done
echo "== done"
