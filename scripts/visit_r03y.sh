#!/bin/bash
# round 3, visit y: the config-4 pass, kernel by kernel
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03y; mkdir -p $out
rm -rf /tmp/prof_c4m
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4m -o c4 -- python $GRAFT_REPO_ROOT/scripts/probe_c4_prof.py > $GRAFT_REPO_ROOT/$out/probe.log 2>&1)
db=$(find /tmp/prof_c4m -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/c4m_kernel_stats.csv && python scripts/rocpd_summary.py seq "$db" 24 > $out/c4m_sequence.csv
grep "fad::\|rocclr" $out/c4m_kernel_stats.csv | cut -c1-130; cat $out/c4m_sequence.csv | cut -c1-110; tail -2 $out/probe.log
echo "== done"
