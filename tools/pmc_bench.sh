#!/bin/bash
# HBM traffic of the bench's dominant kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
# over a short run of the same bench command; per-launch means are written by pmc_traffic.py.
# usage (on the GPU box): bash tools/pmc_bench.sh <outdir>
set -u
OUT=$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 280 rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 $GRAFT_REPO_ROOT/bench.py --num-of-step 2 --no-cpu-baseline > "$OUT/$c.log" 2>&1 || { echo "pass $c failed"; tail -3 "$OUT/$c.log"; exit 1; }
done
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py "$OUT" > "$OUT/traffic.json" && cat "$OUT/traffic.json"
