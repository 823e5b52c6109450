# ROIPool forward: two builds of the library (detect-to-track_amd/lib, lib_ab), bench_ops timings side by side
for lib in lib lib_ab; do
  echo "== $lib"; D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/$lib/libd2t_ops.so python3 bench_ops.py --iters 50 --full 0 2>/dev/null | grep '"roipool' | grep fwd | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ',d['shape'],d['dir'],round(d['us'],2))"
done
