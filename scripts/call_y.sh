export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
timeout 300 python scripts/probe_guard.py 2>&1 | grep "D="
