export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15 | grep -v "^RCCL"
