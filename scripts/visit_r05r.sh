#!/bin/bash
# round 5, visit r: do idle extra streams slow the realistic loop?  + the stall probe with Python's collector passes logged
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05r}; out=gpurun_out/$tag; mkdir -p $out
for n in 0 3 8 24; do for m in attached detached; do EXTRA_STREAMS=$n timeout 200 python scripts/probe_realistic.py $m pipelined 2>&1 | grep "^extra" | cut -c1-110; done; done | tee $out/extra_streams.txt
GPU_MAX_HW_QUEUES=8 EXTRA_STREAMS=24 timeout 200 python scripts/probe_realistic.py attached pipelined 2>&1 | grep "^extra" | cut -c1-110 | sed 's/^/GPU_MAX_HW_QUEUES=8 /' | tee -a $out/extra_streams.txt
timeout 300 python scripts/probe_stall.py 400 2>&1 | grep -E "median|collector" | cut -c1-400 | tee $out/probe_stall.txt
echo "== done"
