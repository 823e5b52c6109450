#!/bin/bash
# A/B of two builds of libd2t_ops.so in ONE GPU session (box-to-box spread is larger than most single changes), inside
# bench.py's alternating fwd/bwd step -- the place the metric runs the kernels.
#   make -C detect-to-track_amd/csrc -j4 OUT=../lib_ab EXTRA=-DD2T_EXP_...=0     # the other build
#   gpurun -- 'bash tools/ab.sh lib lib_ab [rounds] [extra bench.py args]'
A=${1:-lib}; B=${2:-lib_ab}; R=${3:-3}; shift 3 2>/dev/null
for i in $(seq $R); do
  for lib in $A $B; do
    echo -n "$lib  "
    D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/$lib/libd2t_ops.so timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --graph 0 --extras 0 --ops 0 "$@" 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', round(d['ms_per_step']*1000,1), 'events', round(d['event_pass']['ms_per_step']*1000,1), [(k['kernel'], round(k['us'],1), round(k.get('us_min',0),1)) for k in d['kernels']])"
  done
done
