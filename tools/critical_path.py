"""per-queue view of one detect call in a rocprofv3 kernel trace: which queue is busy when, and the main queue's kernels
with their durations under overlap.   python tools/critical_path.py <trace csv> [call index from the end, default 1]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows))
def short(k):
    m = re.search(r'(k_\w+(<[\w, ]*>)?)', k)
    return m.group(1) if m else k.split('<')[0].split('(')[0][-40:]
starts = [i for i, e in enumerate(ev) if 'k_state_init' in e[3]]
i0 = starts[-back]; i1 = starts[-back + 1] if back > 1 else len(ev)
call = [e for e in ev[i0:i1]]
# end of the call: k_finish
for j, e in enumerate(call):
    if 'k_finish' in e[3]:
        call = call[:j + 1]; break
t0 = call[0][0]; t1 = call[-1][1]
print(f'call span {(t1 - t0) / 1e6:.2f} ms, {len(call)} kernels')
byq = {}
for s, e, q, k in call:
    byq.setdefault(q, []).append((s, e, short(k)))
for q, lst in byq.items():
    busy = sum(e - s for s, e, _ in lst)
    print(f'queue {q}: {len(lst)} kernels, busy {busy / 1e6:.2f} ms, from {(lst[0][0] - t0) / 1e6:.2f} to {(lst[-1][1] - t0) / 1e6:.2f} ms')
mainq = max(byq, key=lambda q: len(byq[q]))
agg = {}
prev_end = t0; gaps = 0
for s, e, k in byq[mainq]:
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
    if s > prev_end: gaps += s - prev_end
    prev_end = max(prev_end, e)
print(f'main queue {mainq}: idle gaps between its kernels {gaps / 1e6:.2f} ms')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f'   {k:28s} x{v[0]:3d} {v[1] / 1e6:7.3f} ms')
print('--- kernels longer than 0.8 ms, in start order (q = queue)')
for s, e, q, k in call:
    if (e - s) / 1e6 > 0.8:
        print(f'  q{q} {short(k):26s} start {(s - t0) / 1e6:7.2f} dur {(e - s) / 1e6:6.2f}')
