export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python scripts/probe_c4.py 2>&1 | grep -v amdgpu.ids
