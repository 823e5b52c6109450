#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_ps_roipool.py tests/test_tuned_vs_generic_fuzz.py tests/test_graph_capture.py -m gpu -q -x -k "ps or capture" > gpurun_out/pytest_ps.log 2>&1
echo "pytest rc=$?"; tail -n 5 gpurun_out/pytest_ps.log
timeout -k 10 300 python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "detect-to-track_amd"); sys.path.insert(0, ".")
from conftest import ADVERSARIAL_ROIS, random_rois
from oracle import oracle
from detect_to_track.models import _ext
nT, H, W, k = 16, 38, 63, 7
adv = np.asarray(ADVERSARIAL_ROIS, np.float32)
rois = np.concatenate([adv] * 60 + [random_rois(500, 3)], 0)
rois = rois[np.random.default_rng(5).permutation(len(rois))]
gout = np.random.default_rng(6).standard_normal((len(rois), nT, k, k)).astype(np.float32)
t = lambda a: torch.from_numpy(a).cuda()
gin = _ext.ps_roipool_backward(t(gout), t(rois), H, W).cpu().numpy()
o32 = oracle.psroipool_bwd(gout, rois, H, W)
o64 = oracle.psroipool_bwd(gout.astype(np.float64), rois.astype(np.float64), H, W)
print("R", len(rois), "max|g|", np.abs(o64).max(), "err vs o32", np.abs(gin - o32).max(), "err vs o64", np.abs(gin - o64).max(), "o32 vs o64", np.abs(o32 - o64).max())
PY
