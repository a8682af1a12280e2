export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "frechet or thread" 2>&1 | tail -3
