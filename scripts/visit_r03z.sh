#!/bin/bash
# round 3, visit z: the exact products of a batch on 128 x 64 tiles through LDS
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03z; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "songs or song or batched or indiv" > $out/pytest_songs.log 2>&1; echo "pytest rc=$?"; tail -12 $out/pytest_songs.log | cut -c1-300
timeout 600 python scripts/songs_probe.py acc5 2>&1 | grep "^acc" | cut -c1-200
for shape in "512 1100 72" "768 1500 32"; do
  set -- $shape
  rm -rf /tmp/prof_g
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o g -- python $GRAFT_REPO_ROOT/scripts/songs_probe.py gen $1 $2 $3 4 > $GRAFT_REPO_ROOT/$out/probe_$1.log 2>&1)
  db=$(find /tmp/prof_g -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/gen$1_kernel_stats.csv
  echo "shape $shape"; grep "nsf_\|song_" $out/gen$1_kernel_stats.csv | cut -c1-120; grep call $out/probe_$1.log | tail -1
done
echo "== done"
