# scan build: workgroups per channel of k_psroipool_bwd_gather (rows form), bench_ops timings
export D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/lib_knobs/libd2t_ops.so
for gx in 0 1 2 3 5; do
  echo "gx=$gx"; D2T_PS_GATHER_GX=$gx python3 bench_ops.py --iters 30 --full 1 2>/dev/null | grep psroipool | grep bwd | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ',d['shape'],round(d['us'],2))"
done
