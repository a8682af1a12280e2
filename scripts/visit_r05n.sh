#!/bin/bash
# round 5, visit n: which side block of the bench slows the realistic block that follows it (r05i: 4237 / attached 2876 in the full line, 4832 / 4485 alone)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r05n}; out=gpurun_out/$tag; mkdir -p $out
for sk in "none" "streams" "perstep" "repeat,same" "perstep,streams" "repeat,same,perstep,streams"; do
  FAD_BENCH_REALISTIC_BATCHES=6 FAD_BENCH_SKIP=$sk timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $out/b.json 2> $out/b.err
  python - $out/b.json "$sk" <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); x = r.get("realistic") or {}
print("skip", sys.argv[2], "| value", round(r["value"]), "| realistic", {k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in x.items() if k in ("value", "value_with_attached_walk", "value_with_rounded_exact_means", "error")})
PY
done 2>&1 | tee $out/bisect.txt
echo "== done"
