#!/usr/bin/env python3
"""Screen the gfx950 ISA of every kernel for MFMAs whose destination and source-C register ranges
PARTIALLY overlap (vdst != srcC but intersecting).  hipcc (ROCm 7.2) can emit this after unrolling a
loop that carries several MFMA accumulators (rotating-register scheme); the multi-pass MFMA then reads
C registers it has already overwritten, silently corrupting a shifted group of accumulators (seen in
hsm_bwd_passB<64>).  Run by km-unet_amd/build.py after compiling; exits non-zero on a hit."""
import re
import subprocess
import sys

# vdst overlapping srcA / srcB of a multi-pass bf16 / f16 MFMA: the destination is written while the operand is still being read
PAT_AB = re.compile(r"(v_mfma_f32_16x16x32_(?:bf16|f16)|v_mfma_f32_32x32x16_(?:bf16|f16))\s+([av])\[(\d+):(\d+)\],\s*([av])\[(\d+):(\d+)\],\s*([av])\[(\d+):(\d+)\]")
PAT = re.compile(r"v_mfma_\S+\s+([av])\[(\d+):(\d+)\],\s*\S+,\s*\S+,\s*([av])\[(\d+):(\d+)\]")


WR = re.compile(r"^(v_accvgpr_write_b32|v_accvgpr_mov_b32|v_mov_b32_e32|v_mov_b32)\s+([av])(\d+),")
WINDOW = 8   # wait states hipcc itself leaves (s_nop 7) between a 16x16x4 f32 MFMA and a VALU write to its SrcC
# The bf16 / f16 16x16x32 MFMA is half as long (16 cycles): where hipcc's hazard recognizer DOES see this WAR it leaves 6 wait
# states (one MFMA + s_nop 4); only a write closer than that is the unpadded pattern this screen exists for.
WINDOW_BY_OP = {"v_mfma_f32_16x16x32_bf16": 6, "v_mfma_f32_16x16x32_f16": 6}


def scan(asm_text):
    hits, kernel = [], "?"
    lines = asm_text.splitlines()
    pending = []   # (file, lo, hi, remaining, text) SrcC ranges of recently issued MFMAs
    for ln, line in enumerate(lines, 1):
        s = line.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if s.startswith(".type") and "@function" in s:
                pass
            else:
                continue
        w = WR.match(s)
        if w:
            f, r = w.group(2), int(w.group(3))
            for (pf, lo, hi, _, text, dst) in pending:
                if pf == f and lo <= r <= hi and not (dst[0] == f and dst[1] <= r <= dst[2]):
                    hits.append((kernel, ln, "WAR on SrcC: '%s' right after '%s'" % (s, text)))
        n = re.match(r"s_nop\s+(\d+)", s)
        states = int(n.group(1)) + 1 if n else 1
        pending = [(a, b, c, d - states, e, g) for (a, b, c, d, e, g) in pending if d > states]
        if s.startswith(".type") and "@function" in s:
            kernel = s.split()[1].split(",")[0]
        mab = PAT_AB.search(s)
        if mab:
            fd, d0, d1 = mab.group(2), int(mab.group(3)), int(mab.group(4))
            for (fo, o0, o1) in ((mab.group(5), int(mab.group(6)), int(mab.group(7))), (mab.group(8), int(mab.group(9)), int(mab.group(10)))):
                if fd == fo and not (d1 < o0 or o1 < d0):
                    hits.append((kernel, ln, "vdst overlaps srcA/srcB: " + s))
        m = PAT.search(s)
        if m:
            fd, d0, d1, fc, c0, c1 = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4), int(m.group(5)), int(m.group(6))
            if fd == fc and (d0, d1) != (c0, c1) and not (d1 < c0 or c1 < d0):
                hits.append((kernel, ln, s))
            pending.append((fc, c0, c1, WINDOW_BY_OP.get(s.split()[0], WINDOW), s, (fd, d0, d1)))
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
