export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
for m in plain spin other sleep plain; do timeout 300 python scripts/probe_cold.py $m 2>/dev/null | tail -1; done | tee gpurun_out/r6k_probe_cold.txt
