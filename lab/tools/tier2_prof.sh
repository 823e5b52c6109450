#!/bin/bash
# rocprofv3 kernel stats of tools/envelope_cost.py (every op inside and just outside the tuned envelope): the second kernel tier by kernel
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/t2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/envelope_cost.py > $out/run.log 2>&1 </dev/null
f=$(find $out -name "*kernel_stats.csv" | head -1); grep -E '^"Name"|d2t::' "$f" > $out/tier2_kernel_stats.csv; cut -c1-150 $out/tier2_kernel_stats.csv | head -50
find $out -name "*.csv" -size +1M -delete; find $out -name "*agent_info*" -delete
