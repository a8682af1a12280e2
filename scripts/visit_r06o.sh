export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
for v in "$@"; do
  timeout 300 python scripts/probe_chain16.py --lib scripts/probes/bin/libfad_$v.so 2>/dev/null | tail -1
  timeout 300 python scripts/probe_chain16.py --lib scripts/probes/bin/libfad_$v.so --pairs 20 2>/dev/null | tail -1
done
v=$1
rm -rf $out/prof_chain
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_chain -o b -- python $GRAFT_REPO_ROOT/scripts/probe_chain16.py --lib $GRAFT_REPO_ROOT/scripts/probes/bin/libfad_$v.so > /dev/null 2>&1)
db=$(find $out/prof_chain -name "*.db" | head -1)
python scripts/rocpd_summary.py stats "$db" | grep -E "kernel|nsf" | cut -c1-140
rm -rf $out/prof_chain
