export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('value',round(d['value'],1),'kernel_ms',round(d['roofline']['kernel_ms'],5),'frac',round(d['roofline']['frac'],4));print({k:(v if not isinstance(v,dict) else '...') for k,v in d['extra']['per_song_config5_shape'].items()})"
