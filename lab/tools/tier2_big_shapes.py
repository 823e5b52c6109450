import sys, torch, warnings
sys.path.insert(0, "detect-to-track_amd"); sys.path.insert(0, ".")
from detect_to_track.models import _ext
warnings.simplefilter("ignore")
dev="cuda:0"
for (B,C,H,W,d,s) in [(1,5,130,257,5,1),(1,3,3,1000,14,1),(2,17,64,64,2,2),(1,9,301,5,9,1),(1,4,5,4,14,1),(1,33,40,300,12,3),(1,2,1,4,0,1)]:
    g=torch.Generator().manual_seed(B+C)
    f0=(torch.rand(B,C,H,W,generator=g)-0.4).to(dev); f1=(torch.rand(B,C,H,W,generator=g)-0.4).to(dev)
    go=torch.randn(B,H,W,2*d+1,2*d+1,generator=g).to(dev)
    a=_ext.pointwise_correlation_forward(f0,f1,d,s,0); b=_ext.pointwise_correlation_forward(f0,f1,d,s,1)
    a0,a1=_ext.pointwise_correlation_backward(go,f0,f1,d,s,0); b0,b1=_ext.pointwise_correlation_backward(go,f0,f1,d,s,1)
    print((B,C,H,W,d,s), torch.equal(a,b), torch.equal(a0,b0), torch.equal(a1,b1), flush=True)
from bench_ops import random_rois
import numpy as np
for (R,C,H,W,k) in [(50,3,200,255,9),(2000,4,30,40,4),(64,300,100,100,16),(40,2,255,10,3)]:
    rois=torch.from_numpy(random_rois(R,1)).to(dev); fm=torch.randn(C,H,W,device=dev); go=torch.randn(R,C,k,k,device=dev)
    a=_ext.roipool_forward(fm,rois,k,0); b=_ext.roipool_forward(fm,rois,k,1)
    print("roi",(R,C,H,W,k), torch.equal(a.isnan(),b.isnan()), float((torch.nan_to_num(a)-torch.nan_to_num(b)).abs().max()), torch.equal(_ext.roipool_backward(go,rois,H,W,0), _ext.roipool_backward(go,rois,H,W,1)), flush=True)
for (R,nT,H,W,k) in [(50,3,200,255,9),(2000,4,30,40,4),(64,31,100,100,10)]:
    rois=torch.from_numpy(random_rois(R,1)).to(dev); go=torch.randn(R,nT,k,k,device=dev)
    print("ps",(R,nT,H,W,k), torch.equal(_ext.ps_roipool_backward(go,rois,H,W,0), _ext.ps_roipool_backward(go,rois,H,W,1)), flush=True)
