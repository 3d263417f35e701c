#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sq.sh <outdir-under-gpurun_out> <binary> [args...]
# SQ counter passes (8 SQ slots per pass -- MI355X_MICROARCH.md), kernel trace only; the program itself follows `--`
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
BIN=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU" \
         "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/p$i" -o pmc -- "$BIN" "$@" > "$OUT/p$i.log" 2>&1
  echo "rocprofv3 pass $i exit $?" >> "$OUT/p$i.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as f:
    for k, d in acc.items():
        f.write(k + '\n')
        for c, v in sorted(d.items()):
            f.write(f'  {c:28s} {sum(v) / len(v):16.0f}  (n={len(v)})\n')
print(open(out + '/summary.txt').read())
PY
