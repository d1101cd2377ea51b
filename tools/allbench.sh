set -e
cd $GRAFT_REPO_ROOT
for w in bow cfg1 video kmeans knn logo; do
  timeout -k 10 200 python bench.py --workload $w --no-comm --cpu-sample 0 --steps 10 > gpurun_out/r2w_$w.log 2> gpurun_out/r2w_$w.err
done
timeout -k 10 200 python bench.py --no-comm --steps 10 --cpu-sample 100000 > gpurun_out/r2w_bowp.log 2> gpurun_out/r2w_bowp.err
