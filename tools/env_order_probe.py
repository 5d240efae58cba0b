"""Does MIOpen honour MIOPEN_DEBUG_* variables that are set after `import torch`?  argv[1] = before | after | none"""
import os, sys
mode = sys.argv[1]
V = "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC"
os.environ.pop(V, None)
if mode == "before":
    os.environ[V] = "0"
import torch
if mode == "after":
    os.environ[V] = "0"
os.environ["KMU_NO_ENV_POLICY"] = "1"
sys.argv = [sys.argv[0]]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("mode", mode, "env now", os.environ.get(V))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "smoke_bisect.py")).read())
