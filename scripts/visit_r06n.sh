export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -q -x --tb=short -p no:cacheprovider -k "segmented or online or file_mean or config4 or fused or shifted" > gpurun_out/r6n_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r6n_pytest.log
python - <<'PY'
import sys, json
sys.path.insert(0, '.')
import torch
import bench
from fadtk_amd import hip
r = bench.extra_c4(torch, hip, torch.device('cuda', 0), 0)
print(json.dumps({k: r[k] for k in ('ms', 'frac_of_8TBps', 'tile_kernel_ms_per_update', 'tile_kernel_frac_of_8TBps', 'one_update_of_all_files', 'with_reference_order_file_means')}, indent=None)[:1200])
PY
