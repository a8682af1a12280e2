#!/bin/bash
# round 3, visit q: where the per-song routes lose accuracy (exact diagonal of the float16 covariances)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03q; mkdir -p $out
for w in acc4 acc5; do
  for env in "" "FAD_SONG_COV16=0" "FAD_SONG_FAST=0" "FAD_SONG_COV16=0 FAD_SONG_RES=0" "FAD_SONG_STATS16=0"; do
    env $env timeout 600 python scripts/songs_probe.py $w 2>&1 | grep "^acc" | cut -c1-220
  done
done | tee $out/accuracy.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "songs or song or batched or indiv or gram" > $out/pytest_songs.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_songs.log | cut -c1-300
echo "== done"
