#!/bin/bash
# round 5, eighth visit: the realistic batch with three batches in flight (as bench.py keeps them) -- kernel statistics and the timeline
# (which kernels of the two streams overlap) for the detached walk, the attached walk and no walk.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05h}
out=gpurun_out/$tag; mkdir -p $out
for mode in detached attached off; do
  rm -rf /tmp/prof_x
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $GRAFT_REPO_ROOT/scripts/probe_realistic.py $mode pipelined > $GRAFT_REPO_ROOT/$out/realistic_$mode.txt 2>&1)
  db=$(find /tmp/prof_x -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/realistic_${mode}_kernel_stats.csv && python scripts/rocpd_timeline.py "$db" 260 > $out/realistic_${mode}_timeline.csv
  grep "^mode" $out/realistic_$mode.txt; grep "fad::" $out/realistic_${mode}_kernel_stats.csv | cut -c1-110 | head -16
done
timeout 120 python scripts/probe_realistic.py detached pipelined | grep "^mode"
timeout 120 python scripts/probe_realistic.py off pipelined | grep "^mode"
echo "== done"
