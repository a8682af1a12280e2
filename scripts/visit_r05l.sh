#!/bin/bash
# round 5, visit l: the walk with one wait per 16 rows + global loads + scalar job fields; staging area kept (host path); fuzz test fixed
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05l}; out=gpurun_out/$tag; mkdir -p $out
timeout 300 python scripts/probe_runsum.py > $out/probe_runsum.txt 2>&1; grep "sets=\|differ\|two updates" $out/probe_runsum.txt
FAD_MOMENTS_RUNSUM_COLS=32 timeout 300 python scripts/probe_runsum.py 2>&1 | grep "side=1 sets=" | sed 's/^/cols=32 /' | tee -a $out/probe_runsum.txt
for m in detached attached off; do timeout 200 python scripts/probe_realistic.py $m pipelined 2>&1 | grep "^mode" | tee -a $out/realistic.txt; done
for kb in 0 12288 24576; do FAD_H2D_PIECE_KB=$kb timeout 200 python scripts/probe_host_pieces.py 2>&1 | grep "PIECE_KB\|update" ; done | tee $out/probe_host_pieces.txt
timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "wide_chain or hostile or numpys_own_mean or shifted or golden_g3 or segmented or file_mean or individual or songs" > $out/pytest_new.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_new.log | tail -12 | cut -c1-400
echo "== done"
