export TMPDIR=/tmp
B=scripts/probes/bin
for v in new nocap; do
  echo "== $v  2 x [100000 x 512]"; timeout 120 $B/t256_$v 512 100000 2>&1 | grep -v "^  sl= [842]" | head -12
done
