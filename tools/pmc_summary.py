#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel (hand-written kernels only) into one JSON.

    pmc_summary.py out.json <label>=<counter_collection.csv> [...]

Per kernel and counter: the mean over dispatches.  FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; `hbm_bytes` adds
them (FETCH_SIZE x 2 only where --fetch-x2 is given: MI355X_MICROARCH.md's gfx950 correction for 16 B/lane streaming reads)."""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"\(anonymous namespace\)::(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def main():
    out, args = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith("--")]
    x2 = "--fetch-x2" in sys.argv
    table = collections.defaultdict(lambda: collections.defaultdict(list))
    for a in args:
        _, path = a.split("=", 1)
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k:
                table[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in sorted(table.items()):
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d["dispatches"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes"] = (d["FETCH_SIZE"] * (2 if x2 else 1) + d["WRITE_SIZE"]) * 1024
        if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if c in d:
                    d[c + "/WAVE_CYCLES"] = d[c] / d["SQ_WAVE_CYCLES"]
            if d.get("SQ_ACTIVE_INST_LDS"):
                d["LDS_BANK_CONFLICT/ACTIVE_INST_LDS"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_ACTIVE_INST_LDS"]
        res[k] = d
    json.dump({"kernels": res}, open(out, "w"), indent=1)
    for k, d in res.items():
        print(k, {c: (round(v, 3) if isinstance(v, float) else v) for c, v in d.items() if "/" in c or c in ("hbm_bytes", "dispatches")})


if __name__ == "__main__":
    main()
