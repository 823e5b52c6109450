mkdir -p gpurun_out/r6c
python tests/dp_equivalence_main.py --mode single --out /tmp/s1.pt > gpurun_out/r6c/s1.txt 2>&1
python tests/dp_equivalence_main.py --mode single --out /tmp/s2.pt > gpurun_out/r6c/s2.txt 2>&1
python - <<'PY' > gpurun_out/r6c/single_vs_single.txt 2>&1
import torch
a,b=torch.load('/tmp/s1.pt'),torch.load('/tmp/s2.pt')
rows=[]
for k in a['grads']:
    w=a['grads'][k]; g=b['grads'][k]
    rows.append((float((g-w).abs().max())/max(float(w.abs().max()),1e-30),k,float(w.abs().max())))
for r in sorted(rows,reverse=True)[:10]: print(r)
print('loss', a['loss_mean'], b['loss_mean'])
PY
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tests/dp_equivalence_main.py --mode ranks --ref /tmp/s1.pt > gpurun_out/r6c/ranks.txt 2>&1
tail -12 gpurun_out/r6c/single_vs_single.txt; grep dp_equivalence gpurun_out/r6c/ranks.txt; tail -2 gpurun_out/r6c/ranks.txt
