# same-box A/B of library builds on whole bench.py steps: tools/ab_bench.sh <workload> <lib.so>...
set -e
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
w=$1; shift
for rep in 1 2 3; do
  for lib in default "$@"; do
    if [ "$lib" = default ]; then unset IC_AMD_LIBRARY; else export IC_AMD_LIBRARY=$root/$lib; fi
    line=$(timeout -k 10 200 python $root/bench.py --workload $w --no-comm --cpu-sample 0 --steps 400 --warmup 5 2>/dev/null | tail -n 1)
    echo "$lib $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("ms_per_step %.4f kernel %.4f frac %.3f" % (j["ms_per_step"], j["roofline"]["avg_kernel_ms"], j["roofline"]["frac"]))')"
  done
done
