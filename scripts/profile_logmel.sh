export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; rm -rf $out/p4
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/p4 -o b -- python $GRAFT_REPO_ROOT/scripts/probe_logmel.py > $out/r06r_probe_logmel.txt 2>/dev/null)
db=$(find $out/p4 -name "*.db" | head -1)
python scripts/rocpd_summary.py stats "$db" | grep -E "^kernel|logmel|resample" > $out/r06r_logmel_kernel_stats.csv
cat $out/r06r_probe_logmel.txt $out/r06r_logmel_kernel_stats.csv | cut -c1-160
rm -rf $out/p4
