#!/bin/bash
# The third roof of bench.py's line: L1 / texture-address cache-line rate.
#   gpurun --timeout 900 -- 'bash tools/ta_roof.sh <tag>'
# 1. lab/csrc/ta_lab: every CU streams 16-byte pieces; wall time per wave-instruction for four lane->address maps, and the
#    same launches under rocprofv3 --pmc (TCP_TOTAL_CACHE_ACCESSES = cache-line tag accesses, TA busy / buffer cycles,
#    GRBM_GUI_ACTIVE for the clock): peak = the line rate of the whole-line stream (map 3).
# 2. bench.py under the same counters: lines per launch of k_corr_fwd_seg / k_corr_bwd_strip8.
# 3. tools/ta_roof.py writes profiles/ta_roof.json from both.
tag=${1:-r04}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/ta_$tag; mkdir -p $out
lab=$root/lab/csrc/ta_lab
[ -x $lab ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o $lab $root/lab/csrc/ta_lab.hip || exit 1
$lab > $out/ta_lab_wall.txt 2>&1 || exit 1
cat $out/ta_lab_wall.txt
cd /tmp && export TMPDIR=/tmp
C1="TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE"
C2="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUFFER_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum"
run() { name=$1; shift; mkdir -p $out/$name; timeout -k 10 300 rocprofv3 "$@" > $out/$name/run.log 2>&1 </dev/null; echo "$name rc=$?"; }
run lab_pmc1 --kernel-trace --pmc $C1 --output-format csv -d $out/lab_pmc1 -- $lab &&
run lab_pmc2 --kernel-trace --pmc $C2 --output-format csv -d $out/lab_pmc2 -- $lab &&
BS="python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph 0 --extras 0 --ops 0" &&
run bench_pmc1 --kernel-trace --pmc $C1 --output-format csv -d $out/bench_pmc1 -- $BS &&
run bench_pmc2 --kernel-trace --pmc $C2 --output-format csv -d $out/bench_pmc2 -- $BS
cd $root
python3 tools/ta_roof.py $out > $out/ta_roof.json && cat $out/ta_roof.json
find $out -name "*.csv" -size +2M -delete; find $out -name "*agent_info*" -delete
