#!/bin/bash
# bit 1 no S reloads, 2 no tile stores, 4 no G requests, 8 no ring writes, 16 no MFMAs
for abl in 0 1 2 4 12 16 7 15; do
  echo -n "ABL=$abl  "; D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/lib_lab/libd2t_ops.so D2T_BF_ABL=$abl timeout -k 10 120 python lab/tools/r3_bf_abl.py 2>/dev/null | tail -1
done
