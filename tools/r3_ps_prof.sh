#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/prof_ps; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for shape in "3000 4 38 75" "3000 31 38 75"; do
  d=$out/$(echo $shape | tr ' ' '_'); mkdir -p $d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/tools/r3_ps_one.py $shape > $d/run.log 2>&1 </dev/null
  echo "== $shape"; python3 - <<PY
import csv,glob
f=glob.glob("$d/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'd2t' in r['Name']: print(f"{r['Name'].split('(')[0][-50:]:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
find $out -name "*.csv" -size +1M -delete
