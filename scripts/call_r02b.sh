export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "moments or fuzz" > gpurun_out/pytest_moments.log 2>&1; echo "pytest moments rc=$?"; tail -15 gpurun_out/pytest_moments.log
timeout 600 python scripts/probe_moments.py > gpurun_out/probe_moments.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/probe_moments.txt
