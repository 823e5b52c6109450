#!/bin/bash
# timing ablations of the 8-pixel-strip backward (lab builds: -DW8_ABL=mask, results are wrong): bench.py --impl 6 with each library
for a in ${ABLS:-"" 1 3 4 8 16 31}; do
  lib=lib${a:+_abl$a}
  echo -n "abl ${a:-0}  "
  D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/$lib/libd2t_ops.so timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --graph 0 --extras 0 --ops 0 --impl ${IMPL:-6} 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print([(k['kernel'], round(k['us'],1), round(k.get('us_min',0),1)) for k in d['kernels']])"
done
