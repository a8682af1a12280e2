export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider -k "indexed or score_inf or tile256" > gpurun_out/r6m_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r6m_pytest.log
python - <<'PY'
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
import bench, fadtk_amd
a, b = bench.make_sets(torch, torch.device('cuda', 0), 1, 0)
print(json.dumps(bench.extra_score_inf(fadtk_amd, a.cpu().numpy(), b.cpu().numpy()), indent=None)[:1500])
PY
