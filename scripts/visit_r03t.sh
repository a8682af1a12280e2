#!/bin/bash
# round 3, visit t: D = 384 on the fast chains; where nsf_big's time goes (same workgroup count, a third of the k range)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03t; mkdir -p $out
timeout 600 tests/native/nsfast_check 384 > $out/nsfast_check.txt 2>&1; echo "nsfast rc=$?"; grep -E "FAIL|passed|FAILED" $out/nsfast_check.txt | head
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "nine or songs_full" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log | cut -c1-300
for shape in "256 600 288" "512 1100 72" "768 1500 32"; do
  set -- $shape
  rm -rf /tmp/prof_g
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o g -- python $GRAFT_REPO_ROOT/scripts/songs_probe.py gen $1 $2 $3 4 > $GRAFT_REPO_ROOT/$out/probe_$1.log 2>&1)
  db=$(find /tmp/prof_g -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/gen$1_kernel_stats.csv
  echo "shape $shape"; grep "nsf_\|song_" $out/gen$1_kernel_stats.csv | cut -c1-120; grep call $out/probe_$1.log | tail -1
done
echo "== done"
