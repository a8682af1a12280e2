export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out
bash scripts/gpu_round.sh full 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | grep -v '^{"metric' | tail -32
echo "== kernel sequence"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/prof2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --inflight 1 --no-cpu-baseline --no-extras > /dev/null 2>&1)
db=$(find $out/prof2 -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py seq "$db" > $out/kernel_seq.csv; tail -14 $out/kernel_seq.csv | cut -c1-110
echo "== pmc tile"
bash scripts/pmc_tile.sh 2>&1 | tail -3
rm -rf $out/prof3 && mkdir -p $out/prof3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof3 -o sg -- python $GRAFT_REPO_ROOT/scripts/probe_songs_general.py 2>&1 | grep "songs of")
db=$(find $out/prof3 -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/sg_kernel_stats.csv; head -6 $out/sg_kernel_stats.csv | cut -c1-130
find $out/prof $out/prof2 $out/prof3 $out/pmc -name "*.db" -delete 2>/dev/null
