#!/bin/bash
# round 3, visit s: chain kernels with 21-25 KB of LDS (co-resident with two tile-kernel workgroups)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03s; mkdir -p $out
timeout 600 tests/native/nsfast_check 512 256 1024 > $out/nsfast_check.txt 2>&1; echo "nsfast rc=$?"; grep -E "FAIL|passed|FAILED" $out/nsfast_check.txt | head
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_native.py -m gpu -q --tb=short -p no:cacheprovider -x -k "frechet or songs or song or native or nine" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log | cut -c1-300
for mode in "" "--single-stream"; do
  timeout 600 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-extras $mode > $out/b.json 2> $out/b.err
  python - "$mode" <<'PY'
import json, sys
r = json.load(open("gpurun_out/r03s/b.json"))
al = r["roofline"].get("alone") or {}
print("%-20s value %.0f  ms/step %.4f  repeat-median %.0f  tile kernel %.4f ms  frac %.3f  alone %s  frechet %.4f" % (sys.argv[1] or "lane streams", r["value"], r["ms_per_step"], r["value_repeat_blocks"]["median"], r["roofline"]["kernel_ms"], r["roofline"]["frac"], al.get("kernel_ms"), r["breakdown_ms"]["frechet"]))
PY
done | tee $out/streams.txt
echo "== done"
