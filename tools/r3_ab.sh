#!/bin/bash
# Round 3: parity of the correlation kernels, then an A/B of the backward kernels in one GPU session
# (impl 0 = default dispatch, impl 3 = the 16-wave strip kernel).   gpurun -- 'bash tools/r3_ab.sh'
out=gpurun_out/r3_ab; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_pointwise_correlation.py tests/test_correlation_tracker.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
for i in 1 2 3; do
  for impl in 3 0; do
    echo -n "impl=$impl  "
    timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --impl $impl 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1000,1), [(k['kernel'], round(k['us'],1), round(k.get('us_min',0),1)) for k in d['kernels']])"
  done
done | tee $out/ab.txt
