export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "moments or fuzz or config4 or fused or online or encodec_48k" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/pytest_gpu.log
timeout 600 python scripts/probe_c4.py 2>&1 | grep -v amdgpu.ids
