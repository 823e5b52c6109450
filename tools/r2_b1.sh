#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_pointwise_correlation.py tests/test_tuned_vs_generic_fuzz.py tests/test_correlation_tracker.py tests/test_graph_capture.py -m gpu -q -x > gpurun_out/pytest_b1.log 2>&1
echo "pytest rc=$?"; tail -n 4 gpurun_out/pytest_b1.log
timeout -k 10 300 python bench_ops.py --iters 30 2>/dev/null | grep -E '"corr"|tracker' > gpurun_out/ops_b1.log; cut -c1-150 gpurun_out/ops_b1.log
