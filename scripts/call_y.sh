export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
timeout 300 python bench.py --steps 100 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('value',round(d['value'],1),'ms/step',round(d['ms_per_step'],4),'kernel_ms',round(d['roofline']['kernel_ms'],5),'frac',round(d['roofline']['frac'],4),'fad',d['fad'],d['breakdown_ms']);print({k:v for k,v in d['extra']['c4_moments'].items() if k!='includes'})"
