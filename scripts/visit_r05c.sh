#!/bin/bash
# round 5, third visit (three-set LDS prefetch in the walk, per-song running sums in a kernel of their own, start scale 3.0):
# (scripts/probe_runsum.py, side stream on / off), decaying pairs (scripts/probe_illcond.py), the GPU suite, the bench line.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05c}
out=gpurun_out/$tag; mkdir -p $out
timeout 300 tests/native/nsfast_check > $out/nsfast_check.txt 2>&1; echo "nsfast_check rc=$?"; grep -c " ok$" $out/nsfast_check.txt; grep -E "FAIL|passed|FAILED|mu0" $out/nsfast_check.txt | head -30
timeout 300 python scripts/probe_runsum.py > $out/probe_runsum_side.txt 2>&1; echo "probe_runsum rc=$?"; grep -v amdgpu.ids $out/probe_runsum_side.txt | tail -12
FAD_MOMENTS_RUNSUM_SIDE=0 timeout 300 python scripts/probe_runsum.py > $out/probe_runsum_inline.txt 2>&1; grep "sets=" $out/probe_runsum_inline.txt
timeout 300 python scripts/probe_illcond.py > $out/probe_illcond.txt 2>&1; echo "probe_illcond rc=$?"; grep -E "spectrum|Error|error" $out/probe_illcond.txt | cut -c1-200
FAD_FRECHET_WIDE=0 timeout 300 python scripts/probe_illcond.py > $out/probe_illcond_narrow.txt 2>&1; grep "spectrum" $out/probe_illcond_narrow.txt | cut -c1-200
if [ "$2" != "notests" ]; then
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $out/pytest_gpu.log | tail -40 | cut -c1-300
fi
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - $out <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1] + "/bench.json"))
    print("value", round(r["value"]), "ms_per_step", round(r["ms_per_step"], 4), "roofline", r["roofline"]["kernel"], round(r["roofline"]["kernel_ms"], 4), round(r["roofline"]["frac"], 3), "parity", r.get("parity_rel_err_vs_cpu"))
    for k, v in r.get("extra", {}).items():
        if isinstance(v, dict):
            print(" ", k, {kk: vv for kk, vv in v.items() if kk in ("ms", "songs_per_s", "ok", "scores_per_s", "ms_batched_device_route", "frac_of_8TBps", "error", "k^-0.5", "k^-1", "k^-2")})
except Exception as e:
    print("bench line unreadable:", e)
PY
tail -5 $out/bench.err | cut -c1-300
echo "== done"
