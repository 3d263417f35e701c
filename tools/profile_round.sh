#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <outdir-under-gpurun_out>
# the round's evidence in one go: kernel trace + stats of bench.py, chains serial and overlapped; occupancy view of the
# overlapped trace; SQ / GRBM counters of the serial run (rocprofv3 --pmc, kernel trace only, the program right behind `--`)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --frames 1026 --steps 2 --warmup 1"
cd /tmp && export TMPDIR=/tmp
CPE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/serial" -o trace -- python3 $B > "$OUT/serial.log" 2>&1
echo "serial exit $?" >> "$OUT/serial.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/overlap" -o trace -- python3 $B > "$OUT/overlap.log" 2>&1
echo "overlap exit $?" >> "$OUT/overlap.log"
CPE_SERIAL=1 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o pmc -- python3 $B > "$OUT/pmc_sq.log" 2>&1
echo "pmc exit $?" >> "$OUT/pmc_sq.log"
cd "$GRAFT_REPO_ROOT"
python3 tools/stats_table.py $(find "$OUT/serial" -name '*kernel_stats.csv' | head -1) > "$OUT/kernel_table_serial.txt" 2>&1
python3 tools/occupancy.py $(find "$OUT/overlap" -name '*kernel_trace.csv' | head -1) > "$OUT/occupancy_overlap.txt" 2>&1
python3 tools/occupancy.py $(find "$OUT/serial" -name '*kernel_trace.csv' | head -1) > "$OUT/occupancy_serial.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in glob.glob(out + '/pmc_sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r'(k_\w+)', r['Kernel_Name'])
        if not m: continue
        acc[m.group(1)][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cnt[m.group(1)] += 1
with open(out + '/pmc_sq_summary.csv', 'w') as f:
    f.write('kernel,launches,gpu_cycles,cu_busy_frac,waves_per_cu,valu_active_frac_of_simd_cycles,wave_cycles_issuing,wave_cycles_waiting,wave_cycles_issue_stalled\n')
    rows = []
    for k, d in acc.items():
        cyc = d['GRBM_GUI_ACTIVE'] / 8.0                     # the counter sums the 8 XCDs
        if cyc <= 0: continue
        wc = max(d['SQ_WAVE_CYCLES'], 1.0)
        rows.append((cyc, k, cnt[k], d['SQ_BUSY_CU_CYCLES'] / (cyc * 256), 4 * d['SQ_WAVE_CYCLES'] / (cyc * 256),
                     4 * d['SQ_ACTIVE_INST_VALU'] / (cyc * 1024), d['SQ_ACTIVE_INST_ANY'] / wc, d['SQ_WAIT_ANY'] / wc, d['SQ_WAIT_INST_ANY'] / wc))
    for cyc, k, n, busy, wpc, valu, a, w, s in sorted(rows, reverse=True):
        f.write(f'{k},{n},{cyc:.0f},{busy:.3f},{wpc:.2f},{valu:.3f},{a:.3f},{w:.3f},{s:.3f}\n')
print(open(out + '/pmc_sq_summary.csv').read()[:3000])
PY
head -30 "$OUT/occupancy_overlap.txt"
# HBM traffic per kernel: separate FETCH_SIZE / WRITE_SIZE passes (tools/pmc.sh form) around tools/time_detect.py 32 (64 images per launch)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  CPE_SERIAL=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_hbm/$c" -o pmc -- python3 "$GRAFT_REPO_ROOT/tools/time_detect.py" 32 > "$OUT/pmc_hbm_$c.log" 2>&1
  echo "rocprofv3 $c exit $?" >> "$OUT/pmc_hbm_$c.log"
done
cd "$GRAFT_REPO_ROOT"
python3 tools/pmc_summary.py "$OUT/pmc_hbm" 64 > "$OUT/pmc_summary.csv" 2>&1
head -12 "$OUT/pmc_summary.csv"
# keep the summaries, drop the bulk (gpurun only copies back 64 MiB)
find "$OUT" \( -name '*kernel_trace.csv' -o -name '*counter_collection.csv' -o -name '*.db' -o -name '*agent_info.csv' \) -delete
du -sh "$OUT"
