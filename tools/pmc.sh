#!/bin/bash
# usage (GPU box, repo root): tools/pmc.sh <outdir-under-gpurun_out> <python script> [args...]
# two separate counter passes (TCC slots: FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md), kernel trace only
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o pmc -- python3 "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$OUT/$c.log" 2>&1
  echo "rocprofv3 $c exit $?" >> "$OUT/$c.log"
done
ls -R "$OUT" | head -20
