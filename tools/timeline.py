"""summarise a rocprofv3 kernel trace (trace_kernel_trace.csv): GPU busy time, per-kernel totals, idle gaps.
   python tools/timeline.py <csv> [skip_fraction]   (skip the first part of the run: warm-up)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + (t1 - t0) * skip
ev = [e for e in ev if e[0] >= cut]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = {}
for s, e, k in ev:
    m = re.search(r'(k_\w+(<[\w, ]*>)?)', k)
    k = m.group(1) if m else k.split('<')[0].split('(')[0][-60:]
    a = tot.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
span = t1 - t0
print(f'span {span/1e6:.2f} ms, GPU busy (union of kernels) {busy/1e6:.2f} ms = {100*busy/span:.1f}%, sum of kernel durations {sum(v[1] for v in tot.values())/1e6:.2f} ms')
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'  {k[:60]:60s} x{v[0]:5d} {v[1]/1e6:9.3f} ms')
