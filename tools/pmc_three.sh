export TMPDIR=/tmp
for WL in C3 C5 C2; do
  echo "== $WL"
  WL=$WL bash tools/pmc_counters.sh 2>&1 | grep -v "^f0\|^g0"
done
