"""Does the HIP runtime accept external event-record nodes in a captured graph (torch refuses them on ROCm)?"""
import ctypes, re, sys
sys.path.insert(0, "detect-to-track_amd")
import torch
from detect_to_track.models import _native
path = None
for line in open("/proc/self/maps"):
    m = re.search(r"(/\S*libamdhip64\.so\S*)", line)
    if m:
        path = m.group(1); break
print("hip runtime:", path)
hip = ctypes.CDLL(path)
x = torch.rand(1 << 24, device="cuda")
evs = [ctypes.c_void_p() for _ in range(3)]
for e in evs:
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), 0) == 0
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = [hip.hipEventRecordWithFlags(evs[0], st, 1)]
    y = x * 2
    rc.append(hip.hipEventRecordWithFlags(evs[1], st, 1))
    z = y + 1
    rc.append(hip.hipEventRecordWithFlags(evs[2], st, 1))
print("record rc:", rc)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
ms = ctypes.c_float()
for a, b in ((0, 1), (1, 2), (0, 2)):
    r = hip.hipEventElapsedTime(ctypes.byref(ms), evs[a], evs[b])
    print(a, b, "rc", r, "ms", ms.value)
