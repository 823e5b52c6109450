#!/bin/bash
# dev helper: correlation tests + short bench on the GPU box (run via gpurun)
timeout -k 10 600 python -m pytest tests/test_pointwise_correlation.py -m gpu -x -q > gpurun_out/pytest_corr.log 2>&1
tail -n 2 gpurun_out/pytest_corr.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_dev.log 2>&1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_dev.log").read().strip().splitlines()[-1])
print(round(d["value"], 2), "Gvox/s", round(d["ms_per_step"] * 1e3, 1), "us/step",
      [(k["kernel"], round(k["us"], 1), round(k["mfma"]["achieved"], 1), "TF") for k in d["kernels"]])
PY
