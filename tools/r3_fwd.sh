#!/bin/bash
# Round 3: correlation parity (all kernels incl. the channel-split forward), then bench_ops correlation rows.
out=gpurun_out/r3_fwd; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_pointwise_correlation.py tests/test_correlation_tracker.py tests/test_graph_capture.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout -k 10 600 python bench_ops.py --iters 30 2>/dev/null | grep -E '"corr"|tracker' | tee $out/ops.jsonl | cut -c1-200
