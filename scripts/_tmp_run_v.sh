export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=$GRAFT_REPO_ROOT/gpurun_out
for p in 16 20; do
rm -rf $out/prof_chain
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_chain -o b -- python $GRAFT_REPO_ROOT/scripts/probe_chain16.py --pairs $p > /dev/null 2>&1)
db=$(find $out/prof_chain -name "*.db" | head -1)
echo "== $p pairs, new"; python scripts/rocpd_summary.py stats "$db" | grep -E "nsf" | cut -c1-110
rm -rf $out/prof_chain
done
for p in 4 8 16 20 24; do echo "pairs $p new: $(python scripts/probe_chain16.py --pairs $p | tail -1)";  echo "pairs $p base: $(python scripts/probe_chain16.py --pairs $p --lib scripts/probes/bin/libfad_base.so | tail -1)"; done
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "frechet or chain or song or pairs or multi or declined or decaying" 2>&1 | tail -3
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('parity_rel_err_vs_golden_g7'))"; done
