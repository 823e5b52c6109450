import sys, torch
sys.path.insert(0, "detect-to-track_amd"); sys.path.insert(0, ".")
from detect_to_track.models import _ext
from bench_ops import timed
import warnings; warnings.simplefilter("ignore")
dev="cuda:0"
for (B,C,H,W,d,s) in [(8,256,38,63,7,1),(1,256,38,63,7,1),(8,256,38,63,4,1),(8,256,38,63,12,1)]:
    f0, f1 = torch.rand(B,C,H,W,device=dev), torch.rand(B,C,H,W,device=dev)
    print((B,C,H,W,d,s), round(timed(lambda i: _ext.pointwise_correlation_forward(f0,f1,d,s), 10, 1),1), flush=True)
