#!/bin/bash
# Everything a round's final profiles/<tag>_* set is made from.  Run on the GPU box:  gpurun --timeout 1200 -- 'bash tools/profile_round.sh <tag>'
#   1. kernel durations (rocprofv3 --kernel-trace --stats) of bench.py (the metric workload) and bench_ops.py (every op / shape)
#   2. separate --pmc passes of bench.py: HBM bytes (FETCH_SIZE / WRITE_SIZE), L1<->L2 requests, TA / MFMA / LDS / wait counters
#   3. the third roof: tools/ta_roof.sh (lab/csrc/ta_lab under the same counters) -> ta_roof.json
#   4. rocprof HBM bytes of every op / shape / direction of bench_ops.py (tools/ops_pmc.sh), folded into profiles/traffic.json
#   5. the bench line in the driver's form (--steps 20 --warmup 5) and with 200 steps, bench_ops.jsonl, the config-4 step
tag=${1:-r04}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; mkdir -p $out/$name; timeout -k 10 240 rocprofv3 "$@" > $out/$name/run.log 2>&1 </dev/null; echo "$name rc=$?"; }
B="python3 $root/bench.py --steps 1000 --warmup 20 --no-cpu-baseline --graph 0 --extras 0 --ops 0"
O="python3 $root/bench_ops.py --iters 20"
run bench_stats --kernel-trace --stats --output-format csv -d $out/bench_stats -- $B
run ops_stats   --kernel-trace --stats --output-format csv -d $out/ops_stats -- $O
cd $root
bash tools/pmc_bwd.sh 0 $tag > $out/pmc.log 2>&1; tail -3 $out/pmc.log
bash tools/ta_roof.sh $tag > $out/ta_roof.log 2>&1; tail -3 $out/ta_roof.log
bash tools/ops_pmc.sh $tag > $out/ops_pmc.log 2>&1; tail -3 $out/ops_pmc.log
python3 tools/merge_traffic.py $root/gpurun_out/ops_pmc_$tag $root/gpurun_out/pmc_$tag/pmc_summary.json $tag > $out/traffic_merge.log 2>&1; cp profiles/traffic.json $out/traffic.json
for n in bench_stats ops_stats; do f=$(find $out/$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E '^"Name"|d2t::' "$f" > $out/${n}_kernel_stats.csv; done
python3 bench.py --steps 20 --warmup 5 > $out/bench_line_driver_form.json 2> $out/bench_line.err
python3 bench.py > $out/bench_line_200.json 2>> $out/bench_line.err
python3 bench_ops.py --iters 50 > $out/bench_ops.jsonl 2> /dev/null
timeout -k 10 300 python3 bench_model.py --steps 10 --warmup 3 > $out/model_step.json 2> $out/model_step.err
# keep the merged-back payload small: raw traces stay on the box
find $out -name "*.csv" -size +2M -delete; find $out -name "*agent_info*" -delete
cat $out/bench_stats_kernel_stats.csv | cut -c1-160; cat $out/bench_line_driver_form.json | cut -c1-300
