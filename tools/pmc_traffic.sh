#!/bin/bash
# HBM traffic per kernel and step (GPU box): tools/pmc_traffic.sh f32|f16x3
# Two separate counter passes (FETCH_SIZE, WRITE_SIZE) over a 3-step single-stream bench; no other trace domains.
set -e
PREC=${1:-f16x3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_traffic_$PREC
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python3 $ROOT/bench.py --steps 2 --warmup 1 \
    --no-cpu-baseline --no-roofline --no-overlap --no-h2d --sustained-seconds 0 --settle-steps 0 --precision $PREC > $OUT/$C.log 2>&1
done
# 6 steps: (1 warm-up + 2 timed) before the (empty) settle phase and again after it
python3 $ROOT/tools/pmc_traffic.py $OUT 6 $PREC > $OUT/traffic.json
tail -3 $OUT/traffic.json
