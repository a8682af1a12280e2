#!/bin/bash
# The profile passes of a round (rocprofv3 on the GPU box; --kernel-trace with --stats or with --pmc, never other trace domains):
#   1. kernel statistics + timeline of the timed loop (bench.py --timed-only --steps 32)
#   2. FETCH_SIZE / WRITE_SIZE passes (HBM bytes of moments_tile256: moments_traffic.json is made from them)
#   3. two SQ passes over the tile kernel (matrix pipe, waits, LDS, vector memory)
#   4. kernel statistics of the embedding front end (the config-2 end-to-end block: logmel_kernel)
# Summaries land in gpurun_out/<tag>_*; copy what should be judged into profiles/.
tag=${1:-r06z}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
B="python $GRAFT_REPO_ROOT/bench.py --timed-only --steps 20 --warmup 5"
rm -rf $out/p1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/p1 -o b -- $B > $out/${tag}_bench_profiled.json 2> /dev/null); echo "stats rc=$?"
db=$(find $out/p1 -name "*.db" | head -1)
python scripts/rocpd_summary.py stats "$db" > $out/${tag}_kernel_stats.csv
python scripts/rocpd_summary.py seq "$db" 60 > $out/${tag}_kernel_timeline.csv
head -14 $out/${tag}_kernel_stats.csv | cut -c1-150
rm -rf $out/p1
i=0
while read -r name counters; do
  [ -z "$counters" ] && continue
  rm -rf $out/p2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $counters -d $out/p2 -o b -- $B > /dev/null 2>&1); echo "pmc $name rc=$?"
  db=$(find $out/p2 -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py pmcgrid "$db" | grep -E "^kernel|^#|moments_tile256|moments_reduce256" > $out/${tag}_pmc_$name.csv
  [ "$name" = "FETCH_SIZE" ] && python scripts/rocpd_summary.py schema "$db" | head -5
  cat $out/${tag}_pmc_$name.csv | cut -c1-400
  rm -rf $out/p2
done <<'LIST'
FETCH_SIZE FETCH_SIZE
WRITE_SIZE WRITE_SIZE
SQ_pipe SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS
SQ_lds_vmem SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU
LIST
rm -rf $out/p3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/p3 -o b -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import torch, bench, json
from fadtk_amd import hip
print(json.dumps(bench.extra_c2_vggish(torch, hip, torch.device('cuda', 0), 0, n_files=300)))
" > $out/${tag}_c2_profiled.json 2> /dev/null); echo "c2 stats rc=$?"
db=$(find $out/p3 -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/${tag}_c2_vggish_kernel_stats.csv
grep -E "^kernel|logmel|moments" $out/${tag}_c2_vggish_kernel_stats.csv | cut -c1-150
rm -rf $out/p3
echo "== done"
