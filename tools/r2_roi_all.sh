#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_roipool.py tests/test_tuned_vs_generic_fuzz.py tests/test_graph_capture.py tests/test_correlation_tracker.py tests/test_model_graph.py -m gpu -q -x > gpurun_out/pytest_roi.log 2>&1
echo "pytest rc=$?"; tail -n 6 gpurun_out/pytest_roi.log
for m in pixel mfma; do echo "== $m"; D2T_ROI_BWD=$m timeout -k 10 200 python bench_ops.py --iters 30 2>/dev/null | grep '"roipool"' | grep bwd | cut -c1-130; done
#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_roi -- python3 $GRAFT_REPO_ROOT/tools/roi_bwd_only.py > $out/prof_roi.log 2>&1 </dev/null
f=$(find $out/prof_roi -name "*kernel_stats.csv" | head -1); cut -c1-120 "$f" | head -9
find $out/prof_roi -name "*.csv" -size +1M -delete; find $out/prof_roi -name "*agent_info*" -delete
tail -n 2 $out/prof_roi.log
