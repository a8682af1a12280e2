#!/bin/bash
# round 5, sixth visit: the two shapes of the running-sum walk (FAD_MOMENTS_RUNSUM_COLS = 32 | 16) alone (scripts/probe_runsum.py) and in the
# realistic batch (scripts/probe_realistic.py: detached / attached / off) under rocprofv3 --kernel-trace --stats.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05f}
out=gpurun_out/$tag; mkdir -p $out
for cols in 32 16; do
  export FAD_MOMENTS_RUNSUM_COLS=$cols
  timeout 300 python scripts/probe_runsum.py > $out/probe_runsum_cols$cols.txt 2>&1; echo "== cols $cols"; grep "sets=\|differ\|two updates" $out/probe_runsum_cols$cols.txt
  for mode in detached attached; do
    rm -rf /tmp/prof_x
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $GRAFT_REPO_ROOT/scripts/probe_realistic.py $mode > $GRAFT_REPO_ROOT/$out/realistic_${mode}_cols$cols.txt 2>&1)
    db=$(find /tmp/prof_x -name "*.db" | head -1)
    [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/realistic_${mode}_cols${cols}_kernel_stats.csv
    grep "^mode" $out/realistic_${mode}_cols$cols.txt; grep "colsum\|tile256" $out/realistic_${mode}_cols${cols}_kernel_stats.csv | cut -c1-110
  done
done
unset FAD_MOMENTS_RUNSUM_COLS
timeout 300 python scripts/probe_realistic.py off | grep "^mode"
echo "== done"
