export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out; rm -rf $out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/prof_bench.json 2> $out/prof_bench.err); echo "rocprof rc=$?"
db=$(find $out/prof -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/kernel_stats.csv
cat $out/kernel_stats.csv | cut -c1-160
find $out/prof -name "*.db" -size +8M -delete 2>/dev/null
