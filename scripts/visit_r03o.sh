#!/bin/bash
# round 3, visit o: chunked statistics kernel; k-stagger experiment of the 128 x 128 iteration kernel
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03o; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "songs or song or batched or indiv or gram" > $out/pytest_songs.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_songs.log | cut -c1-300
for sg in 0 1 5 7 13; do
  rm -rf /tmp/prof_s
  (cd /tmp && FAD_BIG_STAGGER=$sg timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $GRAFT_REPO_ROOT/scripts/songs_probe.py c5 4 > $GRAFT_REPO_ROOT/$out/probe_c5_$sg.log 2>&1)
  db=$(find /tmp/prof_s -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/c5_stagger${sg}_kernel_stats.csv; echo "stagger $sg"; grep "nsf_big\|song_stats" $out/c5_stagger${sg}_kernel_stats.csv | cut -c1-110
  grep call $out/probe_c5_$sg.log | tail -1
done
echo "== done"
