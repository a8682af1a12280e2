#!/bin/bash
# round 3, final build: kernel statistics of the per-song extras (songs_probe.py) by shape
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03zz; mkdir -p $out
for shape in "c5:768 1500 32" "d512:512 1100 72" "d256:256 600 288" "c4:128 2250 2000" "d384:384 900 128"; do
  tag=${shape%%:*}; set -- ${shape#*:}
  rm -rf /tmp/prof_g
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o g -- python $GRAFT_REPO_ROOT/scripts/songs_probe.py gen $1 $2 $3 4 > $GRAFT_REPO_ROOT/$out/probe_$tag.log 2>&1)
  db=$(find /tmp/prof_g -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/${tag}_kernel_stats.csv
  echo "shape $tag ($*)"; grep "nsf_\|song_" $out/${tag}_kernel_stats.csv | cut -c1-110 | head -8; grep call $out/probe_$tag.log | tail -1
done
echo "== done"
