#!/usr/bin/env python3
"""GPU idle time from a rocprofv3 --kernel-trace CSV: union of the kernels' [start, end) intervals against the span they
cover (kernels on different streams overlap, so the union, not the sum, is the busy time), and where the gaps are.
  usage: tools/gpu_idle.py <..._kernel_trace.csv> [skip_fraction [stop_fraction]]   (analyse that window of the span)"""
import csv
import sys


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    stop = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    ev = []
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
    ev.sort()
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    cut = t0 + int((t1 - t0) * skip)
    end = t0 + int((t1 - t0) * stop)
    ev = [e for e in ev if cut <= e[0] < end]
    busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    gaps, pairs, prev = [], [], ev[0][2]
    for s, e, n in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
            pairs.append((s - cur_e, prev, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev = n
    busy += cur_e - cur_s
    span = cur_e - ev[0][0]
    print(f"kernels {len(ev)}  span {span / 1e6:.2f} ms  busy (union) {busy / 1e6:.2f} ms = {100.0 * busy / span:.1f} %  "
          f"sum of durations {sum(e - s for s, e, _ in ev) / 1e6:.2f} ms")
    gaps.sort(reverse=True)
    tot_gap = sum(g for g, _ in gaps)
    print(f"gaps: {len(gaps)}, total {tot_gap / 1e6:.2f} ms; > 20 us: {sum(1 for g, _ in gaps if g > 20000)} "
          f"({sum(g for g, _ in gaps if g > 20000) / 1e6:.2f} ms); 5-20 us: {sum(1 for g, _ in gaps if 5000 < g <= 20000)} "
          f"({sum(g for g, _ in gaps if 5000 < g <= 20000) / 1e6:.2f} ms); < 5 us: {sum(g for g, _ in gaps if g <= 5000) / 1e6:.2f} ms")
    by = {}
    for g, n in gaps:
        by[n] = by.get(n, 0) + g
    for n, g in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  idle before {n:60s} {g / 1e6:8.2f} ms")
    _pairs(pairs)


def _pairs(pairs):
    by = {}
    for g, a, b in pairs:
        k = (a[:44], b[:44])
        c = by.setdefault(k, [0, 0])
        c[0] += g; c[1] += 1
    print("largest (kernel before -> kernel after) idle totals:")
    for (a, b), (g, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {g / 1e6:7.2f} ms in {c:5d} gaps ({g / c / 1e3:6.1f} us each)  {a}  ->  {b}")


if __name__ == "__main__":
    main()
