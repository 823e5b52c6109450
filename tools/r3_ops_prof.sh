#!/bin/bash
# bench_ops.py under rocprofv3 --kernel-trace --stats: per-kernel durations of every op / shape.   gpurun -- 'bash tools/r3_ops_prof.sh <tag>'
tag=${1:-ops}; root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ops_stats -- python3 $root/bench_ops.py --iters 20 > $out/ops.jsonl 2> $out/run.err </dev/null; echo "rc=$?"
cd $root
f=$(find $out/ops_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E '^"Name"|d2t::' "$f" | sed 's/d2t::tuned:://; s/(anonymous namespace):://' | cut -c1-70,200-400 > $out/ops_kernel_stats.txt; python3 - <<PY
import csv,glob
f=glob.glob("$out/ops_stats/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'd2t' in r['Name']: print(f"{r['Name'].split('(')[0][-60:]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f}")
PY
find $out -name "*.csv" -size +2M -delete; find $out -name "*agent_info*" -delete
grep -E "corr|tracker" $out/ops.jsonl | cut -c1-150
