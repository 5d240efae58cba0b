"""Build libkmunet_hip.so (gfx950) from csrc/*.hip with hipcc -- in-tree, no JIT cache.

    python km-unet_amd/build.py [--force]

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libkmunet_hip.so")
ARCH = "gfx950"
# (source, extra flags).  dysample.hip generates integer gather indices that must be bit-exact
# against the oracle's fp32 op order => no FMA contraction in that TU.
SOURCES = [
    ("api.hip", []),
    ("kan_conv2d.hip", []),
    ("hsmssd.hip", []),
    ("dysample.hip", ["-ffp-contract=off"]),
    ("deform_conv2d.hip", []),
    ("dwconv3x3.hip", []),
    ("bn_blend.hip", []),
    ("qkv_gate.hip", []),
    ("group_norm.hip", []),
    ("pwconv.hip", []),
    ("colsum.hip", []),
    ("gate_mlp.hip", []),
    ("iwp.hip", []),
    ("gauss11.hip", []),
    ("mix3.hip", []),
    ("shift3.hip", []),
    ("hybrid_loss.hip", []),
    ("contingency.hip", []),
    ("conv3x3_x3.hip", []),
    ("resize.hip", []),
    ("triple_norm.hip", []),
    ("dagem.hip", []),
    ("dagem_fused.hip", []),
    ("ffn_fused.hip", []),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libkmunet_hip.so")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "kmunet_hip.h"))
    hipcc = _hipcc()
    objs, rebuilt = [], False
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc, "-c", s, "-o", o] + COMMON + extra
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            rebuilt = True
        objs.append(o)
    if rebuilt or not os.path.exists(LIB):
        _screen_isa(hipcc)
        subprocess.check_call([hipcc, "-shared", "-o", LIB] + objs + ["--offload-arch=" + ARCH, "-fno-gpu-rdc"])
    return LIB


def _screen_isa(hipcc):
    """Refuse to link if any MFMA has partially overlapping vdst / srcC register ranges (a hipcc 7.2
    codegen hazard that silently corrupts accumulators -- tools/check_mfma_overlap.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kmu_isa_check", os.path.join(HERE, "..", "tools", "check_mfma_overlap.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    for src, extra in SOURCES:
        path = os.path.join(CSRC, src)
        if "mfma" not in open(path).read() and not any("mfma" in open(h).read() for h in [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".inc")]):
            continue
        hits = chk.scan(chk.disassemble(path, hipcc, extra))
        if hits:
            raise RuntimeError("MFMA with partially overlapping vdst/srcC in %s: %s" % (src, hits[:3]))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
