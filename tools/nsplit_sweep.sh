#!/bin/bash
# Kernel time of short scans under forced K-splits (tuning aid; results are identical for every split).
cd ${GRAFT_REPO_ROOT:-.}
for ns in 0 2 3 4 5 6 8; do
  echo "== IC_AMD_NSPLIT=$ns"
  IC_AMD_NSPLIT=$ns timeout -k 10 100 python tools/kbench.py 100000x1000 30000x1000 60000x2000 50000x4000 20000x500 || exit 1
done
