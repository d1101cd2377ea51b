#!/bin/bash
# rocprofv3 evidence for every bench workload (kernel trace + stats, then the --pmc passes), and
# the plain bench line of each: tools/profile_all.sh <tag>
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
for w in bow cfg1 hash video kmeans knn logo histknn; do
  bash tools/profile_round.sh ${tag}_$w --workload $w
  echo "profiled $w"
done
for w in bow cfg1 hash video kmeans knn logo histknn; do
  timeout -k 10 200 python bench.py --workload $w --no-comm --cpu-sample 0 > gpurun_out/${tag}_line_$w.json 2> gpurun_out/${tag}_line_$w.err
  echo "bench $w"
done
