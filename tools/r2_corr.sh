#!/bin/bash
# correlation parity tests + A/B bench against detect-to-track_amd/lib_old
timeout -k 10 600 python -m pytest tests/test_pointwise_correlation.py tests/test_tuned_vs_generic_fuzz.py -m gpu -q -k "corr or north or headline or nonfinite or matches" -x > gpurun_out/pytest_corr.log 2>&1
echo "pytest rc=$?"; tail -n 4 gpurun_out/pytest_corr.log
bash tools/ab_bench.sh > gpurun_out/ab.log 2>&1; cat gpurun_out/ab.log

bash tools/r2_abl.sh
