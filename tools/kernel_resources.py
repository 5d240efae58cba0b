#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py km-unet_amd/csrc/hsmssd.hip [name filter]
"""
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(anonymous namespace\)::", "", o).split("(")[0].replace("void ", "") for o in out]


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    extra = ["-ffp-contract=off"] if src.endswith("dysample.hip") else []
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-c", src, "-o", "/dev/null", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc",
                        "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True)
    rows, cur = [], None
    for line in p.stderr.split("\n"):
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:") or t.startswith("Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = demangle([r["name"] for r in rows])
    print("%-70s %5s %5s %5s %6s %6s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "spillV", "scratch", "occ"))
    for r, n in zip(rows, names):
        if flt and flt not in n:
            continue
        print("%-70s %5s %5s %5s %6s %6s %4s" % (n[:70], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"), r.get("VGPRs Spill", "?"),
                                               r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?")))


if __name__ == "__main__":
    main()
