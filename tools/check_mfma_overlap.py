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


# A non-MFMA instruction reading an MFMA's result before the matrix core has written it back.  hipcc pads this itself for the
# builtins (it leaves >= 8 wait states behind a 16x16x32 bf16 MFMA, >= 10 behind the 16x16x4 f32 one) but knows nothing about an
# MFMA inside inline asm: a register copy it schedules right behind such an asm statement reads stale data (seen in
# hsm_bwd_passB_x3: `v_mov_b64 v[48:49], v[104:105]` one wait state after the MFMA that writes v[104:107]).
# thresholds = the smallest distance hipcc itself leaves anywhere in this library's builtin code (8 for both shapes), minus one
READ_WINDOW = {"v_mfma_f32_16x16x32_bf16": 7, "v_mfma_f32_16x16x32_f16": 7, "v_mfma_f32_16x16x4_f32": 7}
MF_DST = re.compile(r"^(v_mfma_\S+)\s+([av])\[(\d+):(\d+)\],")
REG = re.compile(r"\b([av])(\d+)\b|\b([av])\[(\d+):(\d+)\]")
STORES = ("global_store", "ds_write", "buffer_store", "flat_store", "scratch_store", "global_atomic", "ds_add")


def early_readers(asm_text):
    hits, kernel, pend = [], "?", []
    for ln, line in enumerate(asm_text.splitlines(), 1):
        s = line.strip()
        if s.startswith(".type") and "@function" in s:
            kernel, pend = s.split()[1].split(",")[0], []
            continue
        if not s or s[0] in ";." or s.endswith(":"):
            continue
        s = s.split(";")[0].strip()
        m = MF_DST.match(s)
        n = re.match(r"s_nop\s+(\d+)", s)
        states = int(n.group(1)) + 1 if n else 1
        if not m and not n and pend:
            ops = s.split(None, 1)[1] if " " in s else ""
            parts = [q.strip() for q in ops.split(",")]
            srcs = parts if s.startswith(STORES) else parts[1:]
            regs = set()
            for q in srcs:
                for r in REG.finditer(q):
                    if r.group(1):
                        regs.add((r.group(1), int(r.group(2))))
                    else:
                        regs.update((r.group(3), k) for k in range(int(r.group(4)), int(r.group(5)) + 1))
            for (f, lo, hi, age, need, text) in pend:
                if age < need and any(ff == f and lo <= k <= hi for ff, k in regs):
                    hits.append((kernel, ln, "'%s' reads the result of '%s' after %d wait state(s) (needs %d)" % (s, text, age, need)))
        pend = [(f, lo, hi, age + states, need, t) for (f, lo, hi, age, need, t) in pend if age + states < need]
        if m and m.group(1) in READ_WINDOW:
            f, lo, hi = m.group(2), int(m.group(3)), int(m.group(4))
            pend = [q for q in pend if not (q[0] == f and q[1] == lo and q[2] == hi)]
            pend.append((f, lo, hi, 0, READ_WINDOW[m.group(1)], s))
    return hits


def scan(asm_text):
    hits, kernel = early_readers(asm_text), "?"
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
