#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_roi -- python3 $GRAFT_REPO_ROOT/tools/roi_bwd_only.py > $out/prof_roi.log 2>&1 </dev/null
f=$(find $out/prof_roi -name "*kernel_stats.csv" | head -1); cut -c1-120 "$f" | head -9
find $out/prof_roi -name "*.csv" -size +1M -delete; find $out/prof_roi -name "*agent_info*" -delete
tail -n 2 $out/prof_roi.log
