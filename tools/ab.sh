# same-box A/B of two builds of the library: tools/ab.sh <other .so> <kbench cases...>
set -e
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
other=$1; shift
for rep in 1 2 3; do
  echo "== default build"
  timeout -k 10 120 python -u $root/tools/kbench.py "$@" | grep "N="
  echo "== $other"
  IC_AMD_LIBRARY=$root/$other timeout -k 10 120 python -u $root/tools/kbench.py "$@" | grep "N="
done
