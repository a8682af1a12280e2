#!/bin/bash
# round 3, visit h: the steep song at D = 768, kernel breakdown of the two per-song extras, the nt load policy of the tile kernel
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r03h; mkdir -p $out
for m in "1 1" "0 1" "0 0"; do set -- $m; FAD_SONG_FAST=$1 FAD_SONG_SYM=$2 timeout 600 python scripts/songs_probe.py steep 2>&1 | grep -v amdgpu.ids; done | tee $out/steep.txt
for w in c5 c4; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o $w -- python $GRAFT_REPO_ROOT/scripts/songs_probe.py $w 4 > $GRAFT_REPO_ROOT/$out/probe_$w.log 2>&1)
  f=$(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1); cp "$f" $out/${w}_kernel_stats.csv; head -25 "$f" | cut -c1-200
  grep call $out/probe_$w.log
done
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_default.json 2> $out/bench_default.err; python -c "
import json; r=json.load(open('$out/bench_default.json')); print('default', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'])"
FAD_MOMENTS_LOAD_POLICY=nt timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_nt.json 2> $out/bench_nt.err; python -c "
import json; r=json.load(open('$out/bench_nt.json')); print('nt', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'])"
echo "== done"
