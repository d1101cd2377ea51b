#!/bin/bash
# Collects the rocprofv3 evidence for one bench workload on the GPU box:
#   kernel trace + stats, then one --pmc pass per counter group (never combined with tracing).
# usage: tools/profile_round.sh <tag> [bench.py arguments...]
# Output goes to gpurun_out/<tag>/{stats,pmc_sq,pmc_fetch,pmc_write}; summarise afterwards with
# tools/summarize_rocprof.py and tools/pmc_traffic.py.
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
bench="python3 $root/bench.py --cpu-sample 0 --no-comm $*"
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o p -- $bench --steps 5 --warmup 2 > "$out/stats.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_sq" -o p -- $bench --steps 2 --warmup 1 > "$out/pmc_sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o p -- $bench --steps 2 --warmup 1 > "$out/pmc_fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o p -- $bench --steps 2 --warmup 1 > "$out/pmc_write.log" 2>&1
echo "profiled $tag"
