#!/bin/bash
# Sweep of the tail quota (stages per workgroup of the last, partly filled round) on short scans.
# IC_AMD_QUOTA applies the tail decomposition to every row tile.
for q in 2 3 4 6 8 13 16; do
  echo "== IC_AMD_QUOTA=$q"
  IC_AMD_QUOTA=$q timeout -k 10 100 python tools/kbench.py 100000x1000 30000x1000 60000x2000 50000x4000 20000x500 || exit 1
done
