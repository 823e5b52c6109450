#!/bin/bash
# pooling parity tests + op table
timeout -k 10 900 python -m pytest tests/test_roipool.py tests/test_ps_roipool.py tests/test_tuned_vs_generic_fuzz.py tests/test_graph_capture.py -m gpu -q -x -k "not correlation" > gpurun_out/pytest_pool.log 2>&1
echo "pytest rc=$?"; tail -n 30 gpurun_out/pytest_pool.log
timeout -k 10 300 python bench_ops.py --iters 30 2>&1 | grep -v '"corr"' > gpurun_out/ops.log; cat gpurun_out/ops.log
