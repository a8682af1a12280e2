#!/bin/bash
# round 3, visit g: batched fast chain for songs
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03g; mkdir -p $out
timeout 600 tests/native/nsfast_check 512 256 > $out/nsfast_check.txt 2>&1; echo "nsfast rc=$?"; grep -E "FAIL|passed|FAILED" $out/nsfast_check.txt | head
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "songs or song or frechet or gram or symmetric" > $out/pytest_songs.log 2>&1; echo "pytest rc=$?"; tail -25 $out/pytest_songs.log | cut -c1-400
python - <<'PY' 2>&1 | tail -20
import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from fadtk_amd import hip
dev = torch.device("cuda", 0)
for name, fn in (("c5_frames", lambda: bench.extra_c5_frames(torch, hip, dev)), ("c4_songs", lambda: bench.extra_c4_songs(torch, hip, dev))):
    try:
        r = fn()
        print(name, json.dumps({k: v for k, v in r.items() if k not in ("note", "cpu_baseline")}))
    except Exception as e:
        import traceback; traceback.print_exc()
PY
echo "== done"
