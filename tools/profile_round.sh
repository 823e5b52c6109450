#!/bin/bash
# Everything profiles/r02_* is made from.  Run on the GPU box:  gpurun --timeout 1200 -- 'bash tools/profile_round.sh <tag>'
# kernel durations (rocprofv3 --kernel-trace --stats) and separate --pmc passes (HBM bytes, SQ / LDS / L2 counters)
# for bench.py (the metric workload) and bench_ops.py (every op / shape).
tag=${1:-r02}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; mkdir -p $out/$name; timeout -k 10 240 rocprofv3 "$@" > $out/$name/run.log 2>&1 </dev/null; echo "$name rc=$?"; }
B="python3 $root/bench.py --steps 100 --warmup 10 --no-cpu-baseline"
BS="python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
O="python3 $root/bench_ops.py --iters 20"
run bench_stats --kernel-trace --stats --output-format csv -d $out/bench_stats -- $B
run ops_stats   --kernel-trace --stats --output-format csv -d $out/ops_stats -- $O
run bench_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/bench_fetch -- $BS
run bench_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/bench_write -- $BS
run bench_sq1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/bench_sq1 -- $BS
run bench_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/bench_sq2 -- $BS
run bench_l2 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum --output-format csv -d $out/bench_l2 -- $BS
run ops_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/ops_fetch -- $O
run ops_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/ops_write -- $O
run ops_sq --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/ops_sq -- $O
run ops_l2 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum --output-format csv -d $out/ops_l2 -- $O
cd $root
for n in bench_stats ops_stats; do f=$(find $out/$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E '^"Name"|d2t::' "$f" > $out/${n}_kernel_stats.csv; done
python3 tools/pmc_reduce.py $out/bench_fetch $out/bench_write $out/bench_sq1 $out/bench_sq2 $out/bench_l2 > $out/bench_pmc_summary.json
python3 tools/pmc_reduce.py $out/ops_fetch $out/ops_write $out/ops_sq $out/ops_l2 > $out/ops_pmc_summary.json
python3 bench.py > $out/bench_line.json 2> $out/bench_line.err
python3 bench_ops.py --iters 50 > $out/bench_ops.jsonl 2> /dev/null
# keep the merged-back payload small: raw traces stay on the box
find $out -name "*.csv" -size +2M -delete; find $out -name "*agent_info*" -delete
cat $out/bench_stats_kernel_stats.csv | cut -c1-160; head -c 1500 $out/bench_pmc_summary.json; cat $out/bench_line.json | cut -c1-400
