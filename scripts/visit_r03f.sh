#!/bin/bash
# round 3, visit f: i8 kernels with interleaved loads; bench diagnostics (first-block deficit, tile kernel under stream-per-score)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03f; mkdir -p $out
timeout 600 tests/native/nsfast_check 512 768 > $out/nsfast_check.txt 2>&1; echo "nsfast rc=$?"; grep -E "FAIL|passed|FAILED" $out/nsfast_check.txt | head
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "frechet or score_inf" > $out/pytest_frechet.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_frechet.log
show() { python - "$1" <<'PY'
import json, sys
o = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value", round(o["value"]), "repeat", round(o["value_repeat_blocks"]["median"]), "same_pair", round(o["value_same_pair"]["median"]), "per_stream", o["value_stream_per_score"], "breakdown", o["breakdown_ms"], "kernel_ms", o["roofline"]["kernel_ms"], "spread", o["step_ms_spread"])
PY
}
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_a.json 2> $out/bench_a.err; show $out/bench_a.json
FAD_BENCH_PREWARM=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_prewarm.json 2>/dev/null; show $out/bench_prewarm.json
timeout 600 python bench.py --steps 100 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_100.json 2>/dev/null; show $out/bench_100.json
echo "== kernel trace"
rm -rf $out/prof && mkdir -p $out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2> $GRAFT_REPO_ROOT/$out/prof.err)
db=$(find $out/prof -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/kernel_stats.csv; grep "fad::" $out/kernel_stats.csv | cut -c1-120
find $out/prof -name "*.db" -delete 2>/dev/null
echo "== done"
