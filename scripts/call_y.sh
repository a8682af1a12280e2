export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out
rm -rf $out/prof3 && mkdir -p $out/prof3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/prof3 -o c4 -- python $GRAFT_REPO_ROOT/scripts/probe_c4_prof.py 2>&1 | grep "group of")
db=$(find $out/prof3 -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py seq "$db" > $out/c4_seq.csv; tail -34 $out/c4_seq.csv | cut -c1-120
find $out/prof3 -name "*.db" -delete 2>/dev/null
