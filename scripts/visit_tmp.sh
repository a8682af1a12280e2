export TMPDIR=/tmp
rm -rf /tmp/pc4; (cd /tmp && timeout 200 rocprofv3 --kernel-trace -d /tmp/pc4 -o c4 -- python $GRAFT_REPO_ROOT/scripts/probe_c4_tail.py 2>&1 | grep pass)
db=$(find /tmp/pc4 -name "*.db" | head -1); python scripts/rocpd_timeline.py "$db" 22 | tail -16
python scripts/probe_c4_tail.py 2>&1 | grep pass
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q --tb=short -p no:cacheprovider -x -k "moments or config4 or segment or online or file" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | tail -4
