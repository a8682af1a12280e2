export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
for L in "1" "2"; do
  timeout 300 python bench.py --steps 100 --warmup 8 --inflight $L --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('inflight',d['scores_in_flight'],d['lane_streams'],'value',round(d['value'],1),'ms/step',round(d['ms_per_step'],4),'kernel_ms',round(d['roofline']['kernel_ms'],5),'frac',round(d['roofline']['frac'],4),'fad',d['fad'],'iters',d['newton_schulz_iters'],d['breakdown_ms'])"
done
