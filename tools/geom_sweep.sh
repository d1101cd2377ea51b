for g in 1 2; do
  echo "== geometry $g"
  IC_AMD_GEOMETRY=$g timeout -k 10 120 python -u tools/kbench.py 2000000x1000 2000000x2000 2000000x3000 2000000x4000 2000000x6000 1000000x10000 | grep "N="
done
