#!/bin/bash
# round 3, visit a: tests after the advisor fixes + new bench line + host staging variants
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03a; mkdir -p $out
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest_gpu.log
echo "== host path variants"
for cfg in "threads 8 4096" "threads 4 4096" "threads 16 4096" "threads 8 1024" "threads 8 16384" "register 8 4096" "pageable 8 4096"; do
  set -- $cfg
  FAD_H2D_MODE=$1 FAD_H2D_THREADS=$2 FAD_H2D_CHUNK_KB=$3 timeout 300 python scripts/probe_host_path.py 2>&1 | grep "^\[" | tee -a $out/host_path.txt
done
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cat $out/bench.json | head -c 6000; tail -5 $out/bench.err
echo "== done"
