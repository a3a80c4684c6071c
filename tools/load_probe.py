#!/usr/bin/env python3
"""hipModuleLoad of every offline-built code object of the in-tree cache: which ones does the runtime refuse, and why?"""
import ctypes as C, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
hip = C.CDLL([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0])
hip.hipGetErrorString.restype = C.c_char_p
ok = bad = 0
for f in sorted(glob.glob(os.path.join(ROOT, "infiniteexamodels.jl_amd", "kernels", "*.hsaco"))):
    mod = C.c_void_p()
    rc = hip.hipModuleLoad(C.byref(mod), f.encode())
    if rc == 0:
        ok += 1
        hip.hipModuleUnload(mod)
    else:
        bad += 1
        if bad <= 5:
            print(os.path.basename(f), os.path.getsize(f), "rc", rc, hip.hipGetErrorString(rc).decode())
print("loaded", ok, "refused", bad)
props = torch.cuda.get_device_properties(0)
print(props.gcnArchName)
