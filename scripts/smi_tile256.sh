export TMPDIR=/tmp
B=scripts/probes/bin
for v in ${VARIANTS:-new new_MMAONLY new_NOMMA}; do
  T2_SETS=8 T2_REPS=6000 bash scripts/smi_sample.sh "tile256 $v, 8 x [100000 x 512], 6000 launches back to back" $B/t256_$v 512 100000 2>&1 | grep -v "^GPU\[0\]" | cut -c1-220
done
