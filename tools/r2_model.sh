#!/bin/bash
# BASELINE config 4: full-model training step; lines for both input shapes + rocprofv3 kernel stats
out=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_graph.py -m gpu -q -x > $out/pytest_model.log 2>&1
echo "pytest rc=$?"; tail -n 3 $out/pytest_model.log
timeout -k 10 400 python bench_model.py --steps 10 --warmup 2 > $out/model_608x1008.json 2> $out/model_err.log && echo ok1
timeout -k 10 400 python bench_model.py --steps 10 --warmup 2 --height 608 --width 1200 > $out/model_608x1200.json 2>> $out/model_err.log && echo ok2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_model -- python3 $GRAFT_REPO_ROOT/bench_model.py --steps 10 --warmup 2 > $out/prof_model.log 2>&1 </dev/null
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $out/prof_model -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 40 "$f" | cut -c1-200 > $out/model_kernel_stats_top40.csv
find $out/prof_model -name "*.csv" -size +2M -delete; find $out/prof_model -name "*agent_info*" -delete
grep -v MIOpen $out/model_err.log | tail -n 4; cut -c1-400 $out/model_608x1200.json
