export TMPDIR=/tmp
B=scripts/probes/bin
for v in new new_MMAONLY new_NOMMA; do
  T2_SETS=8 T2_REPS=3000 bash scripts/smi_sample.sh "$v 8x[100000x512] 3000 reps" $B/t256_$v 512 100000 2>&1 | grep -v "^GPU\[0\]" | cut -c1-200
done
