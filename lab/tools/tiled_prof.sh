#!/bin/bash
# per-kernel durations of the correlation just outside the tuned envelope (lab/tools/tiled_cost.py, first N shapes)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/tl; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/lab/tools/tiled_cost.py ${1:-1} > $out/run.log 2>&1 </dev/null
f=$(find $out -name "*kernel_stats.csv" | head -1); grep -E '^"Name"|d2t::' "$f" | cut -c1-220
find $out -name "*.csv" -size +1M -delete; find $out -name "*agent_info*" -delete
