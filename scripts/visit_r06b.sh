export TMPDIR=/tmp
mkdir -p gpurun_out
B=scripts/probes/bin
o=gpurun_out/r6b_tile256.txt; : > $o
for rep in 1 2; do
for v in old new prio ilv ilv_prio nocolsum; do
  echo "== $v  8 x [100000 x 512]" >> $o; T2_SETS=8 timeout 120 $B/t256_$v 512 100000 2>&1 | grep -E "sl= 1|fault|failed" >> $o
  echo "== $v  2 x [100000 x 512]" >> $o; timeout 120 $B/t256_$v 512 100000 2>&1 | grep -E "sl= 1|fault|failed" >> $o
done
done
cut -c1-150 $o
