#!/bin/bash
# usage (GPU box, repo root): tools/quick.sh <tag> [pytest -k expression]
# one iteration of the build -> measure loop: the detect parity tests, the bench line, the serial per-kernel table
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p "$OUT"
python -m pytest tests/test_detect_gpu.py tests/test_boundary_gpu.py tests/test_ccl_gpu.py tests/test_plane_gpu.py -m gpu -x -q ${2:+-k "$2"} > "$OUT/tests.log" 2>&1
rc=$?; tail -3 "$OUT/tests.log"
[ $rc -ne 0 ] && exit $rc
python bench.py --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
python -c "import json,sys; d=json.load(open(sys.argv[1])); print('frames/s', round(d['value'],1), 'top5', [(k['kernel'], k['ms']) for k in d['roofline']['top5']])" "$OUT/bench.json"
cd /tmp && export TMPDIR=/tmp
CPE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/serial" -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --frames 1026 --steps 2 --warmup 1 > "$OUT/serial.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 tools/stats_table.py $(find "$OUT/serial" -name '*kernel_stats.csv' | head -1) > "$OUT/kernel_table_serial.txt" 2>&1
find "$OUT" \( -name '*kernel_trace.csv' -o -name '*.db' -o -name '*agent_info.csv' \) -delete
head -24 "$OUT/kernel_table_serial.txt"; tail -1 "$OUT/kernel_table_serial.txt"
