#!/bin/bash
# round 5, visit 6a: the batched float64 take-over of declined pairs -- its test, the multi tests, the decaying-pairs extra with batches of 16
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r06a}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "multi_job or wide_chain or declined or fuzz_frechet or g10 or score_inf" > $out/pytest_multi.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_multi.log | tail -25 | cut -c1-300
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $out/decaying.txt
import sys, json, torch
sys.path.insert(0, '.')
import bench
from fadtk_amd import hip
r = bench.extra_decaying(torch, hip, torch.device("cuda", 0))
for k, v in r.items():
    if isinstance(v, dict): print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("ms", "route", "iterations", "rel_err_vs_oracle", "ms_per_score_in_a_batch_of_16", "batch_route", "batch_iterations", "batch_rel_err_vs_oracle")})
PY
echo "== done"
