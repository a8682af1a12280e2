export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('value',round(d['value'],1),'fad',d['fad'])")
db=$(find $out/prof2 -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" | grep "gemm_f32_kernel<true>\|gemm_f32_kernel<false>" | cut -c1-100
find $out/prof2 -name "*.db" -delete
