#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_ps
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_ps -- python3 $GRAFT_REPO_ROOT/tools/ps_bwd_only.py > $out/prof_ps.log 2>&1 </dev/null
f=$(find $out/prof_ps -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in csv.reader(open(sys.argv[1])): print(r[0][:60].ljust(60), r[1:4])
PY
find $out/prof_ps -name "*.csv" -size +1M -delete; find $out/prof_ps -name "*agent_info*" -delete
grep "psroipool bwd" $out/prof_ps.log
