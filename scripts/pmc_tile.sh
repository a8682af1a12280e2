#!/bin/bash
# PMC passes over the moments tile kernel at config 3 (rocprofv3 --kernel-trace --pmc only: no other trace domains).
# Usage on the GPU box: bash scripts/pmc_tile.sh ; summaries land in gpurun_out/pmc_tile_<pass>.csv
# SQ counters only: passes with TA_* / TCP_* / TD_* counters never returned on this image (killed by the timeout).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  rm -rf $out/pmct_$i
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $counters -d $out/pmct_$i -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1); echo "pass $i rc=$? : $counters"
  db=$(find $out/pmct_$i -name "*.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py pmc "$db" | grep -E "^kernel|moments_tile_h16" > $out/pmc_tile_$i.csv
  cat $out/pmc_tile_$i.csv
  rm -rf $out/pmct_$i
done <<'LIST'
SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F16
LIST
