export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_dist.py tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider -k "pipeline or dist or config2 or batched or batch_driver or vggish or fused or loader" > gpurun_out/r6h_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r6h_pytest.log
python - <<'PY' > gpurun_out/r6h_extras.json 2> gpurun_out/r6h_extras.err
import json, sys, time
sys.path.insert(0, '.')
import torch, numpy as np
import bench
from fadtk_amd import hip
dev = torch.device('cuda', 0)
out = {}
for name, fn in (("c2_vggish_e2e", lambda: bench.extra_c2_vggish(torch, hip, dev, 0)), ("c4_encodec_embed", lambda: bench.extra_c4_encodec(torch, hip, dev, 0))):
    t = time.time()
    try:
        out[name] = fn()
    except Exception as e:
        import traceback; traceback.print_exc()
        out[name] = {"error": repr(e)}
    out[name + "_wall_s"] = time.time() - t
print(json.dumps(out, indent=1))
PY
echo "extras rc=$?"; cat gpurun_out/r6h_extras.json | head -80; tail -20 gpurun_out/r6h_extras.err
