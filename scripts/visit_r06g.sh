export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
for mg in 4 16; do
rm -rf $out/prof_$mg
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_$mg -o b -- python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 32 --warmup 5 --moments-group $mg > $out/r6g_bench_$mg.json 2> $out/r6g_err_$mg.txt); echo "rc=$?"
db=$(find $out/prof_$mg -name "*.db" | head -1)
python scripts/rocpd_summary.py stats "$db" > $out/r6g_kernel_stats_mg$mg.csv
python scripts/rocpd_summary.py seq "$db" 60 > $out/r6g_kernel_seq_mg$mg.csv
head -16 $out/r6g_kernel_stats_mg$mg.csv | cut -c1-150
tail -45 $out/r6g_kernel_seq_mg$mg.csv | cut -c1-120
rm -rf $out/prof_$mg
done
