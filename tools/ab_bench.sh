#!/bin/bash
# A/B two builds of libd2t_ops.so in one GPU session (box-to-box variance is ~5 %):
#   make -C detect-to-track_amd/csrc OUT=../lib_old   # baseline build of the sources to compare against
#   gpurun -- 'bash tools/ab_bench.sh'
for i in 1 2 3; do
  for v in lib_old lib; do
    echo -n "$v  "
    D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/$v/libd2t_ops.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1000,1), [(k['kernel'], round(k['us'],1)) for k in d['kernels']])"
  done
done
