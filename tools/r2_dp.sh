#!/bin/bash
# rehearsal of BASELINE config 5's code path on ONE GPU: two ranks share the card, gradients averaged over gloo
out=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 300 python bench_model.py --steps 5 --warmup 2 > $out/model_1rank.json 2> $out/model_1rank.err; echo "1 rank rc=$?"
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench_model.py --gpus 2 --backend gloo --steps 5 --warmup 2 > $out/model_2rank_gloo.json 2> $out/model_2rank.err; echo "2 ranks rc=$?"
grep -v MIOpen $out/model_2rank.err | tail -n 5
python - <<'PY'
import json,os
o=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/"
for f in ("model_1rank.json","model_2rank_gloo.json"):
    d=json.load(open(o+f)); print(f, d["n_gpus"], round(d["ms_per_step"],2), "ms", round(d["pairs_per_s"],1), "pairs/s", d["finite"], {k:round(v,2) for k,v in d["sections_ms"].items()})
PY
