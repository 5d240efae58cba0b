#!/usr/bin/env python3
"""Development aid: build an alternative libkmunet_hip.so with extra -D flags for ONE translation unit, next to the shipped one.

    python tools/build_variant.py NAME hsmssd.hip -DKMU_PB_TY64=4 ...   ->  km-unet_amd/lib/variants/NAME/libkmunet_hip.so

Select it with KMU_LIB_VARIANT=NAME (read by km-unet_amd/_lib.py; unset = the shipped library).  Used to time tile-shape /
occupancy alternatives of one kernel on the GPU box in a single gpurun call."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-unet_amd"))
import build as B

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
out = os.path.join(B.LIBDIR, "variants", name)
os.makedirs(out, exist_ok=True)
hipcc = B._hipcc()
extra = dict(B.SOURCES)[src]
obj = os.path.join(out, src.replace(".hip", ".o"))
subprocess.check_call([hipcc, "-c", os.path.join(B.CSRC, src), "-o", obj] + B.COMMON + extra + flags)
objs = [obj if s == src else os.path.join(B.LIBDIR, "obj", s.replace(".hip", ".o")) for s, _ in B.SOURCES]
subprocess.check_call([hipcc, "-shared", "-o", os.path.join(out, "libkmunet_hip.so")] + objs + ["--offload-arch=" + B.ARCH, "-fno-gpu-rdc"])
print(os.path.join(out, "libkmunet_hip.so"))
