#!/bin/bash
# round 5, visit m: why the line's realistic block was slower than scripts/probe_realistic.py -- the block alone, un-profiled and under rocprofv3
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05m}; out=gpurun_out/$tag; mkdir -p $out
for nb in 6 12; do FAD_BENCH_REALISTIC_BATCHES=$nb timeout 300 python bench.py --realistic-only --steps 20 --warmup 5 > $out/realistic_only_$nb.json 2> $out/realistic_only_$nb.err; python - $out/realistic_only_$nb.json $nb <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); x = r.get("realistic") or {}
print("batches per block", sys.argv[2], {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in x.items() if k in ("value", "value_with_attached_walk", "value_with_rounded_exact_means", "steps_per_block", "blocks", "error", "latency_ms_blocking")})
PY
done
rm -rf /tmp/prof_real
(cd /tmp && FAD_BENCH_REALISTIC_BATCHES=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_real -o r -- python $GRAFT_REPO_ROOT/bench.py --realistic-only --steps 20 --warmup 5 > /dev/null 2> $GRAFT_REPO_ROOT/$out/prof.err)
db=$(find /tmp/prof_real -name "*.db" | head -1)
if [ -n "$db" ]; then python scripts/rocpd_summary.py stats "$db" > $out/realistic_only_kernel_stats.csv; python scripts/rocpd_timeline.py "$db" 400 > $out/realistic_only_timeline.csv; fi
grep "fad::" $out/realistic_only_kernel_stats.csv | cut -c1-120 | head -12
echo "== done"
