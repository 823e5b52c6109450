#!/bin/bash
# PMC passes of lab/tools/roi_bwd_only.py (ROIPool backward + forward, config 3): L1<->L2 requests, TA busy, MFMA busy, waits, LDS, HBM bytes.
tag=${1:-roi}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/pmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; mkdir -p $out/$name; timeout -k 10 240 rocprofv3 "$@" > $out/$name/run.log 2>&1 </dev/null; echo "$name rc=$?"; }
BS="python3 $root/lab/tools/roi_bwd_only.py"
run l2 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $out/l2 -- $BS &&
run sq1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/sq1 -- $BS &&
run sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq2 -- $BS &&
run fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $BS &&
run write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $BS
cd $root
python3 tools/pmc_reduce.py $out/l2 $out/sq1 $out/sq2 $out/fetch $out/write > $out/pmc_summary.json
find $out -name "*.csv" -size +2M -delete; find $out -name "*agent_info*" -delete
python3 - <<PY
import json
d=json.load(open("$out/pmc_summary.json"))
for k,v in d.items():
    if "roi" in k: print(k, {c:(round(x["mean"]) if isinstance(x,dict) else round(x,1)) for c,x in v.items()})
PY
