#!/bin/bash
# round 3, visit b: the nine-launch Frechet chain -- kernel checks, parity, bench, kernel sequence
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03d; mkdir -p $out
echo "== native kernel checks"
timeout 600 tests/native/nsfast_check 512 256 768 1024 > $out/nsfast_check.txt 2>&1; echo "nsfast rc=$?"; grep -E "FAIL|==|passed|FAILED|error" $out/nsfast_check.txt | head -60
echo "== parity tests of the Frechet routes"
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "frechet or config3 or score_inf" > $out/pytest_frechet.log 2>&1; echo "pytest rc=$?"; tail -25 $out/pytest_frechet.log
echo "== bench (no extras)"
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_fast.json 2> $out/bench_fast.err; echo "rc=$?"; python - <<'PY'
import json
for name in ("bench_fast",):
    try:
        o = json.load(open(f"gpurun_out/r03d/{name}.json"))
        print(name, "value", o["value"], "ms/step", o["ms_per_step"], "breakdown", o["breakdown_ms"], "fad", o["fad"], "iters", o["newton_schulz_iters"], o["ns_converged"], "repeat", o["value_repeat_blocks"]["median"], "frac", o["roofline"]["frac"])
    except Exception as e:
        print(name, "unreadable", e)
PY
tail -3 $out/bench_fast.err
FAD_FRECHET_FAST=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_f32chain.json 2>/dev/null; python - <<'PY'
import json
o = json.load(open("gpurun_out/r03d/bench_f32chain.json"))
print("f32 chain: value", o["value"], "ms/step", o["ms_per_step"], "breakdown", o["breakdown_ms"], "fad", o["fad"], "repeat", o["value_repeat_blocks"]["median"])
PY
echo "== rocprofv3 kernel trace of the bench"
rm -rf $out/prof && mkdir -p $out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2> $GRAFT_REPO_ROOT/$out/prof.err); echo "rocprof rc=$?"
db=$(find $out/prof -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/kernel_stats.csv && python scripts/rocpd_summary.py seq "$db" 60 > $out/kernel_sequence.csv
head -30 $out/kernel_stats.csv; cat $out/kernel_sequence.csv | tail -45
find $out/prof -name "*.db" -size +8M -delete 2>/dev/null
echo "== full gpu test suite"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest_gpu.log
echo "== done"
