#!/bin/bash
# round 3, visit n: the batched iteration on 128 x 128 tiles
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03n; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "songs or song or batched or indiv or gram or frechet" > $out/pytest_songs.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest_songs.log | cut -c1-300
FAD_FAST_TRACE=1 timeout 600 python scripts/songs_probe.py c5 2 2>&1 | grep -v amdgpu.ids | awk 'NR<=4 || /call/' | cut -c1-220 | tee $out/trace_c5.txt
for w in c5 c4; do
  FAD_SONG_BIG=0 timeout 600 python scripts/songs_probe.py $w 4 2>&1 | grep call | tail -2
  rm -rf /tmp/prof_$w
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o $w -- python $GRAFT_REPO_ROOT/scripts/songs_probe.py $w 4 > $GRAFT_REPO_ROOT/$out/probe_$w.log 2>&1)
  db=$(find /tmp/prof_$w -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/${w}_kernel_stats.csv; head -14 $out/${w}_kernel_stats.csv | cut -c1-150
  grep call $out/probe_$w.log
done
echo "== done"
