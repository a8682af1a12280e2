#!/bin/bash
# round 3, visit w: SQ counters of the resident D = 128 kernel (songs_probe.py c4)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=$GRAFT_REPO_ROOT/gpurun_out/r03w; mkdir -p $out
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  rm -rf /tmp/pmcw_$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $counters -d /tmp/pmcw_$i -o b -- python $GRAFT_REPO_ROOT/scripts/songs_probe.py ${PROBE:-c4} 2 > /dev/null 2>&1); echo "pass $i rc=$? : $counters"
  db=$(find /tmp/pmcw_$i -name "*.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py pmc "$db" | grep -E "^kernel|nsf_res128|nsf_big|nsf_i8_big" > $out/pmc_${PROBE:-c4}_$i.csv
  cat $out/pmc_${PROBE:-c4}_$i.csv
done <<'LIST'
SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY
LIST
echo "== done"
