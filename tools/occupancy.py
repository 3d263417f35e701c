#!/usr/bin/env python3
"""CU-time utilisation of a run from a rocprofv3 kernel trace (trace_kernel_trace.csv).

A kernel that runs W workgroups can occupy at most min(1, W / 256) of the chip's 256 CUs (one workgroup per CU is the
least it needs; kernels whose workgroups share a CU occupy less, so this is an UPPER bound on what the kernel uses).
The timeline is swept: at every instant the weights of the kernels in flight are added (capped at 1) and integrated.
  utilisation = integral / span            what share of the CU-seconds of the span had work on them at best
  busy        = union of kernel intervals  what share of the span had any kernel in flight
Per kernel: calls, total ms, mean workgroups, weight, and its share of the weighted time -- the narrow kernels are the
ones with a large 'ms' and a small 'weighted ms'.
    python tools/occupancy.py <kernel trace csv> [skip fraction, default 0.5 = drop warm-up] [end fraction, default 1]
(fractions of the trace's time span: 0.52 0.9 of a `bench.py --steps 1 --warmup 1` trace lies inside the timed step)"""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    endf = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    ev = []
    for r in rows:
        wg = max(1, int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1))
        grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
        nwg = max(1, grid // wg)
        m = re.search(r'(k_\w+)', r['Kernel_Name'])
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), min(1.0, nwg / 256.0), nwg, m.group(1) if m else 'other'))
    ev.sort()
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    cut = t0 + (t1 - t0) * skip
    cut1 = t0 + (t1 - t0) * endf
    ev = [e for e in ev if e[0] >= cut and e[1] <= cut1]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    pts = []
    for s, e, wt, _, _ in ev:
        pts.append((s, wt, 1)); pts.append((e, -wt, -1))
    pts.sort()
    area = busy = 0.0
    cur = 0.0; live = 0; last = pts[0][0]
    for t, dw, dl in pts:
        if t > last:
            area += min(1.0, cur) * (t - last)
            busy += (t - last) if live > 0 else 0
        cur += dw; live += dl; last = t
    span = t1 - t0
    print(f'span {span / 1e6:.2f} ms; a kernel in flight {100 * busy / span:.1f} % of it; CU-time utilisation (upper bound) {100 * area / span:.1f} %')
    agg = {}
    for s, e, wt, nwg, k in ev:
        a = agg.setdefault(k, [0, 0.0, 0.0, 0])
        a[0] += 1; a[1] += (e - s) / 1e6; a[2] += wt * (e - s) / 1e6; a[3] += nwg
    tot = sum(a[1] for a in agg.values()); totw = sum(a[2] for a in agg.values())
    print(f'sum of kernel durations {tot:.1f} ms, weighted by min(1, workgroups / 256): {totw:.1f} ms ({100 * totw / tot:.1f} %)')
    print(f'{"kernel":28s} {"calls":>6s} {"ms":>9s} {"wgs":>8s} {"weighted ms":>12s} {"lost ms":>8s}')
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:16]:
        print(f'{k[:28]:28s} {a[0]:6d} {a[1]:9.2f} {a[3] // a[0]:8d} {a[2]:12.2f} {a[1] - a[2]:8.2f}')


if __name__ == '__main__':
    main()
