export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out; rm -rf $out/prof
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "frechet or config3 or song or score" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/prof_bench.json 2> $out/prof_bench.err); echo "rocprof rc=$?"
db=$(find $out/prof -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/kernel_stats.csv
cat $out/kernel_stats.csv | cut -c1-160 | head -16
python -c "import json;d=json.load(open('$out/prof_bench.json'));print(d['value'],d['ms_per_step'],d['breakdown_ms'],d['roofline']['frac'],d['newton_schulz_iters'],d['ns_converged'])"
find $out/prof -name "*.db" -size +8M -delete 2>/dev/null
