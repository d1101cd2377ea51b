# same-box comparison of several builds of the library on kbench cases:
# tools/ab_many.sh "<lib.so> <lib.so> ..." <kbench cases...>   ("default" = the in-tree build)
set -e
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
libs=$1; shift
for rep in 1 2 3; do
  for lib in $libs; do
    if [ "$lib" = default ]; then unset IC_AMD_LIBRARY; else export IC_AMD_LIBRARY=$root/$lib; fi
    echo "== $lib"
    timeout -k 10 120 python -u $root/tools/kbench.py 500000x500 "$@" | grep "N=" | tail -n +2
  done
done
