#!/bin/bash
# Round 3 acceptance run: the driver's GPU tier (all -m gpu tests, smoke), the default bench line, the config-4 model step.
out=gpurun_out/r3_full; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
timeout -k 10 400 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-330 $out/bench_line.json
timeout -k 10 500 python bench_model.py --steps 5 --warmup 2 > $out/model_step.json 2> $out/model.err; echo "model rc=$?"; tail -3 $out/model.err; cut -c1-1500 $out/model_step.json
