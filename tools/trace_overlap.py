#!/usr/bin/env python3
"""Dev tool: how much of a two-stream train step really overlaps, from a rocprofv3 --kernel-trace CSV of the NORMAL bench
(weight gradients on their own stream, the Dense head's optimizer range on a third).  Steps are delimited by the
optimizer's closing sum_partials_kernel; per step: the span, the busy time of every hardware queue (union of its kernels'
intervals), the time at least one / at least two queues are busy, and the sum of kernel durations.
usage: trace_overlap.py <kernel_trace.csv> [first_step] [n_steps]"""
import collections, csv, sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("sum_partials_kernel")]
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    return tot + (cur_e - cur_s if cur_e is not None else 0)


def at_least(iv, m):
    ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
    depth, last, tot = 0, None, 0
    for t, d in ev:
        if depth >= m and last is not None:
            tot += t - last
        depth += d
        last = t
    return tot


print("%5s %9s %9s | %s | %9s %9s %9s" % ("step", "span ms", "kernels", "busy ms per queue (launches)", ">=1 busy", ">=2 busy", "idle"))
for k in range(k0, k0 + n):
    seg = rows[marks[k] + 1: marks[k + 1] + 1]
    t0 = int(rows[marks[k]]["End_Timestamp"])
    t1 = int(seg[-1]["End_Timestamp"])
    byq = collections.defaultdict(list)
    for r in seg:
        byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    allv = [iv for q in byq.values() for iv in q]
    per_q = "  ".join("q%s %.3f (%d)" % (q, union(v) / 1e6, len(v)) for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])))
    one, two = at_least(allv, 1), at_least(allv, 2)
    print("%5d %9.3f %9.3f | %s | %9.3f %9.3f %9.3f" % (k, (t1 - t0) / 1e6, sum(e - s for s, e in allv) / 1e6, per_q, one / 1e6,
                                                       two / 1e6, (t1 - t0 - one) / 1e6))
