#!/bin/bash
# round 5, visit j: host rows in pieces (A/B by piece size), the new fuzz / hostile-column tests, the stall probe under --hip-trace again
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05j}
out=gpurun_out/$tag; mkdir -p $out
for kb in 0 6144 12288 24576; do FAD_H2D_PIECE_KB=$kb timeout 200 python scripts/probe_host_pieces.py 2>&1 | grep PIECE_KB; done | tee $out/probe_host_pieces.txt
timeout 200 python scripts/probe_host_pieces.py 2>&1 | grep PIECE_KB | tee -a $out/probe_host_pieces.txt
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "wide_chain or hostile or numpys_own_mean or fuzz_frechet or host or golden_g3 or multi_job" > $out/pytest_new.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_new.log | tail -25 | cut -c1-400
rm -rf /tmp/prof_stall
(cd /tmp && timeout 300 rocprofv3 --hip-trace --kernel-trace -d /tmp/prof_stall -o s -- python $GRAFT_REPO_ROOT/scripts/probe_stall.py 300 > $GRAFT_REPO_ROOT/$out/probe_stall.txt 2>&1)
grep -E "median" $out/probe_stall.txt | cut -c1-300
db=$(find /tmp/prof_stall -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_long_calls.py "$db" 5 > $out/stall_long_calls.txt 2>&1; head -30 $out/stall_long_calls.txt | cut -c1-250
echo "== done"
