#!/bin/bash
# round 3, visit r: stream layouts of the timed loop
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03r; mkdir -p $out
for mode in "" "--lane-streams" "--split-streams" "--lane-streams --inflight 4" "--split-streams --inflight 4"; do
  timeout 600 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-extras $mode > $out/b.json 2> $out/b.err
  python - "$mode" <<'PY'
import json, sys
r = json.load(open("gpurun_out/r03r/b.json"))
print("%-34s value %.0f  ms/step %.4f  repeat-median %.0f  tile kernel %.4f ms  frac %.3f" % (sys.argv[1] or "single stream", r["value"], r["ms_per_step"], r["value_repeat_blocks"]["median"], r["roofline"]["kernel_ms"], r["roofline"]["frac"]))
PY
done | tee $out/streams.txt
echo "== done"
