#!/bin/bash
# per-kernel durations of lab/tools/lists_cost.py
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/ll; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/lab/tools/lists_cost.py > $out/run.log 2>&1 </dev/null
f=$(find $out -name "*kernel_stats.csv" | head -1); grep -E '^"Name"|d2t::' "$f" | cut -c1-100,180-300
grep -E "roipool" $out/run.log
find $out -name "*.csv" -size +1M -delete; find $out -name "*agent_info*" -delete
