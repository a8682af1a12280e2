export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out
bash scripts/gpu_round.sh full 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | grep -v '^{"metric' | tail -40
echo "== kernel sequence"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/prof2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --inflight 1 --no-cpu-baseline --no-extras > /dev/null 2>&1)
db=$(find $out/prof2 -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py seq "$db" > $out/kernel_seq.csv; tail -14 $out/kernel_seq.csv | cut -c1-110
echo "== pmc tile"
bash scripts/pmc_tile.sh 2>&1 | tail -4
echo "== c4 profile"
rm -rf $out/prof3 && mkdir -p $out/prof3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof3 -o c4 -- python $GRAFT_REPO_ROOT/scripts/probe_c4_prof.py 2>&1 | grep "group of")
db=$(find $out/prof3 -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/c4_kernel_stats.csv
echo "== c5 profile"
rm -rf $out/prof4 && mkdir -p $out/prof4
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof4 -o c5 -- python $GRAFT_REPO_ROOT/scripts/probe_c5_prof.py 2>&1 | grep "songs")
db=$(find $out/prof4 -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py stats "$db" > $out/c5_kernel_stats.csv
echo "== guard + illcond probes"
timeout 300 python scripts/probe_guard.py 2>&1 | grep "D=" > $out/probe_guard.txt; cat $out/probe_guard.txt
timeout 300 python scripts/probe_illcond.py 2>&1 | grep "D=" > $out/probe_illcond.txt; cat $out/probe_illcond.txt
find $out/prof $out/prof2 $out/prof3 $out/prof4 $out/pmc -name "*.db" -delete 2>/dev/null
