#!/bin/bash
# Round 3: region filter parity, per-call time, per-kernel time.
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r3_reg; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_region_filter.py -m gpu -x -q > $out/pytest.log 2>&1 </dev/null; echo "pytest rc=$?"; tail -3 $out/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $root/tools/r3_regions_time.py > $out/prof.log 2>&1 </dev/null
grep "us per call" $out/prof.log
python3 $root/tools/r3_regions_trace.py $out/prof ; python3 - <<PY
import csv,glob
f=glob.glob("$out/prof/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'region' in r['Name']: print(f"{r['Name'].split('(')[0][-60:]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
find $out -name "*.csv" -size +8M -delete
