#!/usr/bin/env python3
"""Where does a replayed training step's wall time go?  Reads a rocprofv3 --kernel-trace CSV (one row per dispatch with start / end
timestamps and the hardware queue), cuts it into steps at the fused-AdamW launch, and for the last full steps reports
  * wall, union-busy and idle time, the sum of kernel durations and the concurrency (sum / busy);
  * per queue: launches, busy time, and the distribution of the gap between one kernel's end and the next one's start;
  * the idle windows (no kernel running anywhere): count, total, and the kernels that follow the longest ones;
  * the dependent-launch latency seen on the busiest queue: median / p90 gap when the queue's next kernel starts within 30 us.
usage: timeline.py kernel_trace.csv [out.json]"""
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)[:48]


def main():
    path = sys.argv[1]
    rows = []
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        cols = rd.fieldnames
        def pick(*keys):
            for c in cols:
                if all(k.lower() in c.lower() for k in keys):
                    return c
            raise SystemExit("timeline.py: %s has no column matching %s (columns: %s) -- expected a rocprofv3 --kernel-trace CSV"
                             % (sys.argv[1], "+".join(keys), ", ".join(cols)))
        c_name, c_s, c_e, c_q = pick("kernel", "name"), pick("start"), pick("end"), pick("queue")
        for r in rd:
            nm = r[c_name]
            rows.append((int(r[c_s]), int(r[c_e]), r[c_q], "ADAMW" if "FusedAdam" in nm or "fused_adam" in nm.lower() else short(nm)))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[3] == "ADAMW"]
    # steps = rows between consecutive optimizer launches; keep the last 5 complete ones (graph replays of the timed loop are
    # followed by 3 eager instrumented steps, which carry extra event-record gaps: take the steps with the smallest wall)
    steps = []
    for a, b in zip(adam[:-1], adam[1:]):
        seg = rows[a + 1:b + 1]
        if len(seg) > 200:
            steps.append(seg)
    steps.sort(key=lambda seg: max(r[1] for r in seg) - min(r[0] for r in seg))
    if not steps:
        print("no steps found (%d rows, %d optimizer launches)" % (len(rows), len(adam)))
        return
    seg = steps[len(steps) // 4]          # a fast (replayed) step, not the very fastest outlier
    t0, t1 = min(r[0] for r in seg), max(r[1] for r in seg)
    wall = (t1 - t0) / 1e3
    ksum = sum(r[1] - r[0] for r in seg) / 1e3
    # union busy / idle windows
    ev = sorted(seg)
    busy, idle_windows, cur_end = 0, [], ev[0][0]
    for s, e, q, n in ev:
        if s > cur_end:
            idle_windows.append((s - cur_end, n, cur_end - t0))
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    busy /= 1e3
    out = {"launches": len(seg), "wall_us": wall, "busy_us": busy, "idle_us": wall - busy, "kernel_sum_us": ksum,
           "concurrency": ksum / busy if busy else 0, "idle_windows": len(idle_windows)}
    print("step: %d launches, wall %.0f us, some kernel running %.0f us (idle %.0f us in %d windows), kernel sum %.0f us, concurrency %.2f"
          % (len(seg), wall, busy, wall - busy, len(idle_windows), ksum, out["concurrency"]))
    # idle window histogram
    iw = sorted(w[0] / 1e3 for w in idle_windows)
    if iw:
        q = lambda p: iw[min(len(iw) - 1, int(p * len(iw)))]
        print("idle windows: median %.1f us, p90 %.1f us, max %.1f us; <2us: %d, 2-5: %d, 5-10: %d, >10: %d" % (
            q(0.5), q(0.9), iw[-1], sum(w < 2 for w in iw), sum(2 <= w < 5 for w in iw), sum(5 <= w < 10 for w in iw), sum(w >= 10 for w in iw)))
        out["idle_hist"] = {"median": q(0.5), "p90": q(0.9), "max": iw[-1]}
        after = {}
        for w, n, _ in idle_windows:
            a = after.setdefault(n, [0, 0.0])
            a[0] += 1
            a[1] += w / 1e3
        print("idle time by the kernel that ends the window (top 25):")
        for n, (c, tot) in sorted(after.items(), key=lambda kv: -kv[1][1])[:25]:
            print("   %-48s %4d windows %8.1f us" % (n, c, tot))
        out["idle_by_next"] = {n: {"windows": c, "us": tot} for n, (c, tot) in sorted(after.items(), key=lambda kv: -kv[1][1])[:40]}
    # exclusive time: the part of each kernel's duration during which nothing else ran -- the serial skeleton of the step
    pts = sorted([(s, 1, i) for i, (s, e, q, n) in enumerate(ev)] + [(e, -1, i) for i, (s, e, q, n) in enumerate(ev)])
    live, excl, last = set(), {}, pts[0][0]
    depth_time = {}
    for t, kind, i in pts:
        if t > last:
            d = len(live)
            depth_time[d] = depth_time.get(d, 0) + (t - last)
            if d == 1:
                n = ev[next(iter(live))][3]
                a = excl.setdefault(n, [0.0, set()])
                a[0] += (t - last) / 1e3
                a[1].add(next(iter(live)))
        last = t
        if kind == 1:
            live.add(i)
        else:
            live.discard(i)
    print("time by number of kernels in flight: " + ", ".join("%d: %.0f us" % (d, v / 1e3) for d, v in sorted(depth_time.items())))
    out["time_by_depth_us"] = {str(d): v / 1e3 for d, v in sorted(depth_time.items())}
    print("exclusive time (only this kernel running) by kernel, top 40:")
    for n, (tot, ids) in sorted(excl.items(), key=lambda kv: -kv[1][0])[:40]:
        print("   %-48s %4d launches %8.1f us" % (n, len(ids), tot))
    out["exclusive_us"] = {n: {"launches": len(ids), "us": tot} for n, (tot, ids) in sorted(excl.items(), key=lambda kv: -kv[1][0])[:60]}
    # per queue
    queues = {}
    for s, e, q, n in ev:
        queues.setdefault(q, []).append((s, e, n))
    print("queues:")
    out["queues"] = {}
    for q, lst in sorted(queues.items(), key=lambda kv: -len(kv[1])):
        b = sum(e - s for s, e, _ in lst) / 1e3
        gaps = sorted((lst[i + 1][0] - lst[i][1]) / 1e3 for i in range(len(lst) - 1))
        close = [g for g in gaps if g < 30]
        med = close[len(close) // 2] if close else 0
        p90 = close[int(0.9 * len(close))] if close else 0
        neg = sum(g < 0 for g in gaps)
        print("   queue %-6s %5d launches, busy %8.0f us, back-to-back gap median %.1f us p90 %.1f us (of %d gaps < 30 us; %d overlapping)"
              % (q, len(lst), b, med, p90, len(close), neg))
        out["queues"][q] = {"launches": len(lst), "busy_us": b, "gap_median_us": med, "gap_p90_us": p90}
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
    if len(sys.argv) > 3:          # the step's dispatches in start order: offset, duration, queue, name
        with open(sys.argv[3], "w") as f:
            for s_, e_, q_, n_ in ev:
                f.write("%9.1f %7.1f q%s %s\n" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, q_, n_))


if __name__ == "__main__":
    main()
