import ctypes, sys
sys.path.insert(0, "detect-to-track_amd")
import torch
from detect_to_track.models import _native
lib = _native.lib
torch.zeros(1, device="cuda:0")
f = lib.d2t_lab_roipool_fwd_occupancy
f.restype = ctypes.c_int
b, r, s, d = (ctypes.c_int() for _ in range(4))
for thr in (1024, 512, 256):
    rc = f(38, 63, thr, ctypes.byref(b), ctypes.byref(r), ctypes.byref(s), ctypes.byref(d))
    print("threads", thr, "rc", rc, "blocks/CU", b.value, "regs", r.value, "static LDS", s.value, "dynamic LDS", d.value)
p = torch.cuda.get_device_properties(0)
print(p.name, "CUs", p.multi_processor_count, "max threads/CU", p.max_threads_per_multi_processor, "shared/CU", getattr(p, "shared_memory_per_multiprocessor", None), "regs/CU", getattr(p, "regs_per_multiprocessor", None))
