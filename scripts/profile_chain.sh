export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
for v in "$@"; do
rm -rf $out/prof_chain
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_chain -o b -- python $GRAFT_REPO_ROOT/scripts/probe_chain16.py --lib $GRAFT_REPO_ROOT/scripts/probes/bin/libfad_$v.so > /dev/null 2>&1)
db=$(find $out/prof_chain -name "*.db" | head -1)
echo "== $v"; python scripts/rocpd_summary.py stats "$db" | grep -E "nsf" | cut -c1-120
rm -rf $out/prof_chain
done
