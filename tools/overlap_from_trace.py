#!/usr/bin/env python
"""What overlaps what in a two-lane run: reads a rocprofv3 --kernel-trace CSV of `bench.py` (two batches in flight: two queues) and prints, per kernel,
its mean duration alone-equivalent (from a second, --inflight 1 trace if given), its mean duration in the two-lane run, and the share of its run time
during which a kernel of the OTHER lane was running, by the other kernel's name.  Markdown on stdout (profiles/r05/two_lane_overlap.md).
usage: tools/overlap_from_trace.py <two_lane_kernel_trace.csv> [<one_lane_kernel_trace.csv>]"""
import csv, sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "")
    if "rocprim" in n: return "rocprim::radix_sort"
    n = n.split("(")[0].replace("compvhip::", "")
    return n.split("<")[0]


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "compvhip" not in n and "rocprim" not in n:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(n)))
    rows.sort()
    return rows


def main(two, one=None):
    rows = load(two)
    # steady state only: drop the first and the last fifth of the dispatches (warm-up, verification passes)
    n = len(rows)
    rows = rows[n // 5: n - n // 5]
    queues = sorted({q for _, _, q, _ in rows})
    alone = defaultdict(list)
    if one:
        for s, e, q, k in load(one):
            alone[k].append(e - s)
    dur = defaultdict(list)
    over = defaultdict(lambda: defaultdict(float))   # kernel -> other kernel -> ns of overlap
    anyover = defaultdict(float)
    by_q = {q: [r for r in rows if r[2] == q] for q in queues}
    for s, e, q, k in rows:
        dur[k].append(e - s)
        covered = []
        for q2 in queues:
            if q2 == q:
                continue
            for s2, e2, _, k2 in by_q[q2]:
                if e2 <= s:
                    continue
                if s2 >= e:
                    break
                lo, hi = max(s, s2), min(e, e2)
                if hi > lo:
                    over[k][k2] += hi - lo
                    covered.append((lo, hi))
        covered.sort()
        tot, cur_lo, cur_hi = 0, None, None
        for lo, hi in covered:
            if cur_hi is None or lo > cur_hi:
                if cur_hi is not None:
                    tot += cur_hi - cur_lo
                cur_lo, cur_hi = lo, hi
            else:
                cur_hi = max(cur_hi, hi)
        if cur_hi is not None:
            tot += cur_hi - cur_lo
        anyover[k] += tot
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _, _ in rows)
    print("queues in the trace: %s; %d dispatches analysed over %.3f ms; sum of kernel durations / wall = %.2f (1.0 = no overlap, 2.0 = two kernels at all times)\n"
          % (queues, len(rows), span / 1e6, busy / span))
    print("| kernel | launches | us alone (one lane) | us in the two-lane run | stretch | share of its time with an other-lane kernel running | mostly beside |")
    print("|---|---|---|---|---|---|---|")
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        d = sum(dur[k]) / len(dur[k]) / 1e3
        a = (sum(alone[k]) / len(alone[k]) / 1e3) if alone.get(k) else None
        tot = sum(dur[k])
        top = sorted(over[k].items(), key=lambda kv: -kv[1])[:3]
        print("| %s | %d | %s | %.1f | %s | %.2f | %s |" % (k, len(dur[k]), ("%.1f" % a) if a else "-", d, ("%.2f" % (d / a)) if a else "-", anyover[k] / tot,
                                                       ", ".join("%s %.2f" % (k2, v / tot) for k2, v in top)))


if __name__ == "__main__":
    main(*sys.argv[1:3])
