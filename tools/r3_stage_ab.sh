#!/bin/bash
# A/B in one session: bench.py with the shipped library (lib) and with another build (lib_lab).
for i in 1 2 3; do
  for lib in lib lib_lab; do
    echo -n "$lib  "
    D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/$lib/libd2t_ops.so timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1000,1), [(k['kernel'], round(k['us'],1), round(k.get('us_min',0),1)) for k in d['kernels']])"
  done
done
