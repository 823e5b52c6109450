#!/bin/bash
# usage: bash tools/prof.sh <tag> <python script and args...>   -> gpurun_out/prof_<tag>/ + printed kernel stats
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$@" > $out/run.log 2>&1 </dev/null
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $out/kernel_stats.csv; head -12 "$f" | cut -c1-200; else tail -5 $out/run.log; fi
