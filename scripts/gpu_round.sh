#!/bin/bash
# One GPU-box visit: smoke, GPU parity tests, bench line, rocprofv3 kernel stats (+ optional PMC pass).
# Usage (from the repo root, on the GPU box):  bash scripts/gpu_round.sh [quick|full]
# Everything lands under gpurun_out/ (merged back by gpurun); copy what should be judged into profiles/.
mode=${1:-full}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out
mkdir -p $out
echo "== rocminfo" ; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -4
echo "== smoke"
timeout 600 python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $out/smoke.log
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -40 $out/pytest_gpu.log
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cat $out/bench.json; tail -5 $out/bench.err
if [ "$mode" = "full" ]; then
  echo "== rocprofv3 kernel stats"
  rm -rf $out/prof && mkdir -p $out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$out/prof_bench.json 2> $GRAFT_REPO_ROOT/$out/prof_bench.err); echo "rocprof rc=$?"
  find $out/prof -name "*kernel_stats*" | head; f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
  echo "== rocprofv3 pmc (HBM bytes)"
  rm -rf $out/pmc && mkdir -p $out/pmc
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$out/pmc/fetch -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1); echo "pmc fetch rc=$?"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$out/pmc/write -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1); echo "pmc write rc=$?"
  # compact summaries (what gets copied into profiles/), then drop the big traces
  db=$(find $out/prof -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/kernel_stats.csv
  db=$(find $out/pmc/fetch -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py pmc "$db" > $out/pmc_fetch.csv
  db=$(find $out/pmc/write -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py pmc "$db" > $out/pmc_write.csv
  head -14 $out/kernel_stats.csv; grep -E "moments_tile|gemm" $out/pmc_fetch.csv $out/pmc_write.csv
  find $out/prof $out/pmc -name "*.db" -size +8M -delete 2>/dev/null
fi
echo "== done"
