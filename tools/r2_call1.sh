#!/bin/bash
# round-2 call 1: headline golden fixture from the live reference, full GPU suite, A/B bench vs the round-1 build
mkdir -p gpurun_out/golden
python tests/golden/make_golden.py --headline-only gpurun_out/golden > gpurun_out/golden.log 2>&1 || { tail -5 gpurun_out/golden.log; exit 1; }
cp gpurun_out/golden/corr_headline_subsample.npz tests/golden/
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_c1.log 2>&1
echo "pytest rc=$?"; tail -n 15 gpurun_out/pytest_c1.log
bash tools/ab_bench.sh > gpurun_out/ab_c1.log 2>&1
cat gpurun_out/ab_c1.log
