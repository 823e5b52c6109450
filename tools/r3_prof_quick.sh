#!/bin/bash
# Quick counter passes for the metric workload (bench.py): kernel stats + SQ / L2 counters.  gpurun -- 'bash tools/r3_prof_quick.sh <tag> [impl]'
tag=${1:-q}; impl=${2:-0}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; mkdir -p $out/$name; timeout -k 10 240 rocprofv3 "$@" > $out/$name/run.log 2>&1 </dev/null; echo "$name rc=$?"; }
BS="python3 $root/bench.py --steps 30 --warmup 5 --no-cpu-baseline --graph 0 --impl $impl"
run bench_stats --kernel-trace --stats --output-format csv -d $out/bench_stats -- $BS
run bench_sq1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/bench_sq1 -- $BS
run bench_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/bench_sq2 -- $BS
run bench_l2 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum --output-format csv -d $out/bench_l2 -- $BS
cd $root
f=$(find $out/bench_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E '^"Name"|d2t::' "$f" > $out/bench_stats_kernel_stats.csv
python3 tools/pmc_reduce.py $out/bench_sq1 $out/bench_sq2 $out/bench_l2 > $out/bench_pmc_summary.json
find $out -name "*.csv" -size +2M -delete; find $out -name "*agent_info*" -delete
cut -c1-200 $out/bench_stats_kernel_stats.csv; cat $out/bench_pmc_summary.json | head -c 3000
