#!/bin/bash
# kernel stats of the default bench command (graph replay + eager event pass) + the default line; run on the GPU box
tag=${1:-final}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_stats -- python3 $root/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out/run.log 2>&1 </dev/null
echo "rocprof rc=$?"
cd $root
f=$(find $out/bench_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E '^"Name"|d2t::' "$f" > $out/kernel_stats.csv
find $out -name "*.csv" -size +2M -delete; find $out -name "*agent_info*" -delete
python3 bench.py > $out/bench_line.json 2> $out/bench_line.err
cut -c1-200 $out/kernel_stats.csv; cut -c1-300 $out/bench_line.json
