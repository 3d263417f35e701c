#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile.sh <outdir-under-gpurun_out> <python script> [args...]
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$OUT/run.log" 2>&1
echo "rocprofv3 exit $?" >> "$OUT/run.log"
ls -R "$OUT" | head -30
