#!/bin/bash
# round 3, visit h: the steep song at D = 768 / 1024, per-song decisions of the batched chain, kernel breakdown of the two per-song extras
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03h; mkdir -p $out
FAD_SONG_FAST=1 FAD_SONG_SYM=1 timeout 600 python scripts/songs_probe.py steep 2>&1 | grep -v amdgpu.ids | tee $out/steep.txt
FAD_FAST_TRACE=1 timeout 600 python scripts/songs_probe.py c5 2 2>&1 | grep -v amdgpu.ids | head -80 | cut -c1-220 | tee $out/trace_c5.txt
FAD_FAST_TRACE=1 timeout 600 python scripts/songs_probe.py c4 2 2>&1 | grep -v amdgpu.ids | awk 'NR<=6 || /call/' | cut -c1-220 | tee $out/trace_c4.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "songs or song" > $out/pytest_songs.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest_songs.log | cut -c1-300
for w in c5 c4; do
  rm -rf /tmp/prof_$w
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o $w -- python $GRAFT_REPO_ROOT/scripts/songs_probe.py $w 4 > $GRAFT_REPO_ROOT/$out/probe_$w.log 2>&1)
  db=$(find /tmp/prof_$w -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/${w}_kernel_stats.csv; head -16 $out/${w}_kernel_stats.csv | cut -c1-150
  grep call $out/probe_$w.log
done
echo "== done"
