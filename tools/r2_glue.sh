#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_correlation_tracker.py tests/test_pointwise_correlation.py tests/test_tuned_vs_generic_fuzz.py -m gpu -q -x -k "not roipool" > gpurun_out/pytest_glue.log 2>&1
echo "pytest rc=$?"; tail -n 25 gpurun_out/pytest_glue.log
timeout -k 10 300 python bench_ops.py --iters 30 2>&1 | grep -E 'tracker|"corr"' > gpurun_out/ops_glue.log; cat gpurun_out/ops_glue.log
