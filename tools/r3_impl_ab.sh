#!/bin/bash
# A/B in one session: bench.py with the default dispatch and with --impl N (e.g. 4 = bf16x3 backward).
impl=${1:-4}
for i in 1 2 3; do
  for im in 0 $impl; do
    echo -n "impl=$im  "
    timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --impl $im 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1000,1), round(d['value'],2), [(k['kernel'], round(k['us'],1), round(k.get('us_min',0),1)) for k in d['kernels']])"
  done
done
