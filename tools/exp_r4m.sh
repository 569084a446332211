#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4m; mkdir -p $OUT; cd $ROOT
COMMON="--steps 40 --warmup 5 --settle-steps 60 --no-cpu-baseline --no-roofline --no-one-stream --no-h2d --sustained-seconds 0"
python3 bench.py $COMMON > /dev/null 2>&1
for rep in 1 2; do
for q in default 2 4 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  v=$(python3 bench.py $COMMON 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['config']['ids_crc32_rank0'])")
  echo "hw_queues=$q rep=$rep value=$v"
  v=$(MTGV_DET_FORK=0 python3 bench.py $COMMON 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['config']['ids_crc32_rank0'])")
  echo "hw_queues=$q det_fork=0 rep=$rep value=$v"
done; done | tee $OUT/hw_queues.txt
