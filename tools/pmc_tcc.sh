#!/bin/bash
# usage (GPU box, repo root): tools/pmc_tcc.sh <outdir-under-gpurun_out>
# L2 / fabric request counters per kernel (separate passes, kernel trace only) around tools/time_detect.py 32:
# atomics that leave the L2 (they execute at the memory side), read / write requests, L2 hits and misses
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in TCC_EA0_ATOMIC_sum "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  d=$(echo $c | tr ' ' '_')
  CPE_SERIAL=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$d" -o pmc -- python3 "$GRAFT_REPO_ROOT/tools/time_detect.py" 32 > "$OUT/$d.log" 2>&1
  echo "rocprofv3 $c exit $?" >> "$OUT/$d.log"
done
cd "$GRAFT_REPO_ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(fn)):
        m = re.search(r'(k_\w+(?:<\w+>)?)', r['Kernel_Name'])
        if not m: continue
        acc[m.group(1)][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'TCC_EA0_ATOMIC_sum': cnt[m.group(1)] += 1
names = ['TCC_EA0_ATOMIC_sum', 'TCC_EA0_RDREQ_sum', 'TCC_EA0_WRREQ_sum', 'TCC_HIT_sum', 'TCC_MISS_sum']
with open(out + '/pmc_tcc_summary.csv', 'w') as f:
    f.write('kernel,launches,' + ','.join(n + '_per_image' for n in names) + ',images_per_launch\n')
    rows = sorted(acc.items(), key=lambda kv: -kv[1]['TCC_EA0_ATOMIC_sum'])
    for k, d in rows:
        n = max(cnt[k], 1)
        # time_detect.py 32 runs 4 detect calls of 64 images: totals / (4 * 64) = per image
        f.write(k + f',{n},' + ','.join(f'{d[x] / 256.0:.0f}' for x in names) + ',64\n')
print(open(out + '/pmc_tcc_summary.csv').read()[:4000])
PY
find "$OUT" \( -name '*kernel_trace.csv' -o -name '*counter_collection.csv' -o -name '*.db' -o -name '*agent_info.csv' \) -delete
