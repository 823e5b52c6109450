#!/bin/bash
# A/B of two implementation selectors of ONE library in one GPU session, inside bench.py's alternating fwd/bwd step.
#   gpurun -- 'bash tools/impl_ab.sh 7 6 [rounds]'        (7 = strips 4 pixels wide, 6 = strips 8 pixels wide)
A=${1:-7}; B=${2:-6}; R=${3:-3}
for i in $(seq $R); do
  for impl in $A $B; do
    echo -n "impl $impl  "
    timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --graph 0 --extras 0 --ops 0 --impl $impl 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', round(d['ms_per_step']*1000,1), 'events', round(d['event_pass']['ms_per_step']*1000,1), [(k['kernel'], round(k['us'],1), round(k.get('us_min',0),1)) for k in d['kernels']])"
  done
done
