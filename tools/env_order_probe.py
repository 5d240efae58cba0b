"""Does MIOpen honour MIOPEN_DEBUG_* variables set after `import torch`?  Uses a switch with an unmistakable effect:
MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0 sends the 64->16 3x3 weight gradient at 128x128 to a ~18 ms naive kernel.
argv[1] = before | after | never"""
import os, sys, time
mode = sys.argv[1]
V = "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM"
os.environ.pop(V, None)
if mode == "before":
    os.environ[V] = "0"
import torch
if mode == "after":
    os.environ[V] = "0"
import torch.nn.functional as F
x = torch.randn(8, 64, 128, 128, device="cuda", requires_grad=True)
w = torch.randn(16, 64, 3, 3, device="cuda", requires_grad=True)
for _ in range(3):
    y = F.conv2d(x, w, padding=1); y.sum().backward()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    y = F.conv2d(x, w, padding=1); y.sum().backward()
torch.cuda.synchronize()
print("mode %-6s conv fwd+bwd %.2f ms" % (mode, (time.perf_counter() - t) / 5 * 1e3))
