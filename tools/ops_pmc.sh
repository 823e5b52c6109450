#!/bin/bash
# rocprof-reported HBM bytes of EVERY op / shape / direction of bench_ops.py (north_star: "rocprof-reported achieved HBM GB/s").
#   gpurun --timeout 900 -- 'bash tools/ops_pmc.sh <tag>'
# Two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: the MI355X guide's recipe; they do not fit one pass) of
# `bench_ops.py --pmc-markers`, which launches a marker kernel (k_corr_mask) in front of every timed series and records
# (label, calls); tools/ops_pmc_reduce.py cuts the dispatch stream at the markers and writes
#   gpurun_out/ops_pmc_<tag>/ops_pmc_summary.json   per label: FETCH_SIZE / WRITE_SIZE (KB) per call, kernels per call, HBM bytes
#   gpurun_out/ops_pmc_<tag>/traffic_ops.json       the "ops" section of profiles/traffic.json (bench_ops.py reads it)
tag=${1:-r05}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/ops_pmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; mkdir -p $out/$name; timeout -k 10 400 rocprofv3 "$@" > $out/$name/run.log 2>&1 </dev/null; echo "$name rc=$?"; }
O="python3 $root/bench_ops.py --iters 5"
run fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $O --pmc-markers $out/fetch/labels.jsonl &&
run write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $O --pmc-markers $out/write/labels.jsonl
cd $root
python3 tools/ops_pmc_reduce.py $out/fetch $out/write $out
find $out -name "*.csv" -size +2M -delete; find $out -name "*agent_info*" -delete
python3 - <<PY
import json
d = json.load(open("$out/ops_pmc_summary.json"))
for k, v in d.items():
    print(f"{k:55s} kernels/call {v['kernels_per_call']:5.1f}  FETCH {v['fetch_kb']:9.0f} KB  WRITE {v['write_kb']:9.0f} KB  HBM {v['hbm_bytes']/1e6:8.2f} MB")
PY
