#!/usr/bin/env python3
"""Screen the gfx950 ISA of every kernel for MFMAs whose destination and source-C register ranges
PARTIALLY overlap (vdst != srcC but intersecting).  hipcc (ROCm 7.2) can emit this after unrolling a
loop that carries several MFMA accumulators (rotating-register scheme); the multi-pass MFMA then reads
C registers it has already overwritten, silently corrupting a shifted group of accumulators (seen in
hsm_bwd_passB<64>).  Run by km-unet_amd/build.py after compiling; exits non-zero on a hit."""
import re
import subprocess
import sys

PAT = re.compile(r"v_mfma_\S+\s+([av])\[(\d+):(\d+)\],\s*\S+,\s*\S+,\s*([av])\[(\d+):(\d+)\]")


def scan(asm_text):
    hits, kernel = [], "?"
    for ln, line in enumerate(asm_text.splitlines(), 1):
        s = line.strip()
        if s.startswith(".type") and "@function" in s:
            kernel = s.split()[1].split(",")[0]
        m = PAT.search(s)
        if m:
            fd, d0, d1, fc, c0, c1 = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4), int(m.group(5)), int(m.group(6))
            if fd == fc and (d0, d1) != (c0, c1) and not (d1 < c0 or c1 < d0):
                hits.append((kernel, ln, s))
    return hits


def disassemble(src, hipcc="hipcc", extra=()):
    cmd = [hipcc, "-S", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", src, "-o", "-"] + list(extra)
    return subprocess.run(cmd, capture_output=True, text=True, check=True).stdout


if __name__ == "__main__":
    bad = 0
    for src in sys.argv[1:]:
        hits = scan(disassemble(src))
        for k, ln, s in hits:
            print("PARTIAL-OVERLAP MFMA in %s (asm line %d): %s" % (k, ln, s))
        bad += len(hits)
        print("%s: %d mfma partial overlaps" % (src, len(hits)))
    sys.exit(1 if bad else 0)
