#!/bin/bash
# A/B in one session: bench.py with the shipped library (lib) and with another build (lib_lab): the metric (K steps, nothing
# else on the stream), the event pass, the per-kernel event intervals.
for i in 1 2 3; do
  for lib in lib lib_lab; do
    echo -n "$lib  "
    D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/$lib/libd2t_ops.so timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --graph 0 --extras 0 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', round(d['ms_per_step']*1000,1), 'events', round(d['event_pass']['ms_per_step']*1000,1), [(k['kernel'], round(k['us'],1), round(k.get('us_min',0),1)) for k in d['kernels']])"
  done
done
