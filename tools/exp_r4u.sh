#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r4u; mkdir -p $OUT
COMMON="--steps 40 --warmup 5 --settle-steps 60 --no-cpu-baseline --no-roofline --no-one-stream --no-h2d --sustained-seconds 0"
python3 bench.py $COMMON > /dev/null 2>&1
for rep in 1 2 3; do
for v in "MTGV_MATCH_PRIO=0" "MTGV_MATCH_PRIO=-1" "MTGV_MATCH_PRIO=-1 MTGV_DET_FORK=0" "MTGV_MATCH_PRIO=0 MTGV_DET_FORK=0"; do
  r=$(env $v python3 bench.py $COMMON 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['config']['ids_crc32_rank0'])")
  echo "[$v] rep=$rep value: $r"
done; done | tee $OUT/match_prio.txt
