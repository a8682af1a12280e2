#!/bin/bash
# round 3, visit v: D = 128 songs, exact products inside the resident kernel
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03v; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "songs or song or batched or indiv" > $out/pytest_songs.log 2>&1; echo "pytest rc=$?"; tail -12 $out/pytest_songs.log | cut -c1-300
for env in "" "FAD_SONG_RES=1"; do env $env timeout 600 python scripts/songs_probe.py acc4 2>&1 | grep "^acc" | cut -c1-200; done | tee $out/accuracy.txt
FAD_FAST_TRACE=1 timeout 600 python scripts/songs_probe.py c4 2 2>&1 | grep -v amdgpu.ids | awk 'NR<=4 || /call/' | cut -c1-220
rm -rf /tmp/prof_c4
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/scripts/songs_probe.py c4 4 > $GRAFT_REPO_ROOT/$out/probe_c4.log 2>&1)
db=$(find /tmp/prof_c4 -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/c4_kernel_stats.csv; grep "nsf_\|song_" $out/c4_kernel_stats.csv | cut -c1-130
grep call $out/probe_c4.log
echo "== done"
