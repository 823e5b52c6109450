#!/bin/bash
# developer tool: the backward strip kernel with parts of its memory traffic removed (lab/csrc/bwd_stamp_lab built with -DD2T_ABL=n;
# bit 0: no gradOut loads, bit 1: no tile stores, bit 2: no S loads).  Results are wrong by construction; timing only.
for n in 0 1 2 4 3 7; do
  echo "== ablation $n"; timeout -k 10 60 lab/csrc/bwd_stamp_lab_$n | grep -E "per launch|  loop|k-blocks|ring put|barrier wait|tile store|wave total" | head -8
done > gpurun_out/abl.log 2>&1
cat gpurun_out/abl.log
