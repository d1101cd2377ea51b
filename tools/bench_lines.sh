#!/bin/bash
# The plain bench line of every workload (no profiler attached): tools/bench_lines.sh <tag>
# writes gpurun_out/<tag>_line_<workload>.json; the default workload runs as the driver runs it
# (CPU baseline, parity block and the communicating sub-records included).
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
timeout -k 10 400 python bench.py > gpurun_out/${tag}_line_default.json 2> gpurun_out/${tag}_line_default.err
echo "bench default"
for w in bow cfg1 hash video kmeans knn logo histknn; do
  timeout -k 10 200 python bench.py --workload $w --no-comm --cpu-sample 0 > gpurun_out/${tag}_line_$w.json 2> gpurun_out/${tag}_line_$w.err
  echo "bench $w"
done
