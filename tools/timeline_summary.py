#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV of bench.py: per train step (delimited by the fused AdamW kernel) the wall
time, the time with 0 / 1 / 2 / 3+ kernels in flight, per-queue busy time, and the kernels ranked by EXCLUSIVE time (time during
which nothing else runs) -- with parallel graph branches the sum of kernel durations no longer says what bounds the step.

    python3 tools/timeline_summary.py <kernel_trace.csv> [n_fastest_steps] [top]"""
import csv, sys, collections

path = sys.argv[1]
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 5
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
ends = [i for i, r in enumerate(rows) if "FusedAdam" in r[2] or "fused_adam" in r[2].lower()]
if len(ends) < nlast + 1:
    sys.exit("only %d AdamW launches found" % len(ends))
allsteps = [(ends[k] + 1, ends[k + 1]) for k in range(len(ends) - 1)]
# the graph-replay steps are the fastest ones (warm-up, validation and the instrumented steps run eagerly)
allsteps.sort(key=lambda ab: max(r[1] for r in rows[ab[0]:ab[1] + 1]) - rows[ab[0]][0])
steps = allsteps[:nlast]
tot = collections.Counter(); excl = collections.Counter(); cnt = collections.Counter()
conc = collections.Counter(); qbusy = collections.Counter(); wall = 0
for a, b in steps:
    seg = rows[a:b + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    wall += t1 - t0
    ev = []
    for i, (s, e, n, q) in enumerate(seg):
        ev.append((s, 1, i)); ev.append((e, -1, i))
        tot[n] += e - s; cnt[n] += 1; qbusy[q] += e - s
    ev.sort()
    live = set(); last = t0
    for t, d, i in ev:
        if t > last:
            k = len(live)
            conc[min(k, 4)] += t - last
            if k == 1:
                excl[seg[next(iter(live))][2]] += t - last
            last = t
        if d > 0: live.add(i)
        else: live.discard(i)
n = len(steps)
print("steps analysed: %d   wall %.3f ms/step" % (n, wall / n / 1e6))
for k in range(5):
    print("  %s kernels in flight: %7.3f ms/step" % (("%d" % k) if k < 4 else "4+", conc[k] / n / 1e6))
print("per-queue busy time (ms/step): " + "  ".join("q%s=%.2f" % (q, v / n / 1e6) for q, v in sorted(qbusy.items(), key=lambda t: -t[1])))
print("\nkernels by EXCLUSIVE time (ms/step)   [exclusive | total | launches]")
for name, v in excl.most_common(top):
    print("  %7.3f | %7.3f | %5.1f  %s" % (v / n / 1e6, tot[name] / n / 1e6, cnt[name] / n, name[:110]))
