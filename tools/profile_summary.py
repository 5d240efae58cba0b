#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv of bench.py: per-category and top-kernel time per step.
usage: profile_summary.py <kernel_stats.csv> <step-equivalents in the run> [top N]"""
import collections
import csv
import sys


def category(n):
    # everything that is not ATen / a BLAS or MIOpen kernel / a runtime copy comes from libkmunet_hip.so (the CSV truncates long
    # names, so a positive list of kernel names goes stale with every new kernel)
    if "at::" not in n and not n.startswith("Cijk") and "rocclr" not in n and "miopen" not in n.lower() and \
            not any(k in n for k in ("igemm", "batched_transpose", "SubTensor", "naive_conv")):
        return "hand-written HIP"
    if n.startswith("Cijk"):
        return "GEMM (hipBLASLt/rocBLAS)"
    if any(k in n for k in ("igemm", "batched_transpose", "SubTensor", "naive_conv")) or "miopen" in n.lower() or ("Conv" in n and "at::" not in n):
        return "MIOpen"
    if "reduce_kernel" in n:
        return "ATen reduce"
    if "multi_tensor" in n:
        return "ATen foreach / AdamW"
    if "FillFunctor" in n:
        return "ATen fill"
    if "elementwise" in n:
        return "ATen elementwise"
    if "copyBuffer" in n or "fillBuffer" in n:
        return "rocclr copy/fill"
    return "other ATen"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    n = float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        k = category(r["Name"])
        agg[k][0] += int(r["Calls"])
        agg[k][1] += float(r["TotalDurationNs"]) / 1e6
    tot = sum(v[1] for v in agg.values())
    print("kernel time %.2f ms/step, %.0f launches/step" % (tot / n, sum(v[0] for v in agg.values()) / n))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-28s %7.1f launches/step %7.3f ms/step %5.1f%%" % (k, v[0] / n, v[1] / n, 100 * v[1] / tot))
    print()
    for r in rows[:top]:
        print("%-96s %7.1f x %8.1f us = %6.3f ms/step" % (r["Name"][:96].replace("(anonymous namespace)::", ""), int(r["Calls"]) / n,
                                                         float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / n))


if __name__ == "__main__":
    main()
