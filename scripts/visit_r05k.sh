#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05k}; out=gpurun_out/$tag; mkdir -p $out
timeout 300 python scripts/probe_case_d768.py 2>&1 | grep -v amdgpu.ids | tee $out/probe_case_d768.txt | cut -c1-400
echo "== 128-kernel moments"; FAD_MOMENTS_TILE256=0 timeout 300 python scripts/probe_case_d768.py 2>&1 | grep "^(a)\|^(c)\|cov" | cut -c1-300 | tee -a $out/probe_case_d768.txt
if /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/dep_add_rate scripts/probes/dep_add_rate.hip; then /tmp/dep_add_rate | tee $out/dep_add_rate.txt; fi
echo "== done"
