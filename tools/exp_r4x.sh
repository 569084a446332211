#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r4x; mkdir -p $OUT
C="--no-cpu-baseline --no-roofline --no-h2d --sustained-seconds 0 --no-one-stream"
port=29520
for v in "" "MTGV_STREAM_PRIO=enc MTGV_MATCH_PRIO=-1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8 MTGV_STREAM_PRIO=enc MTGV_MATCH_PRIO=-1" "GPU_MAX_HW_QUEUES=8 MTGV_STREAM_PRIO=enc MTGV_MATCH_PRIO=0"; do
  port=$((port+1))
  env $v RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port MTGV_FORCE_COLLECTIVE=1 python3 bench.py --gpus 1 $C 2>/dev/null | grep '^{"metric' > $OUT/line.json
  python3 -c "import json;d=json.loads(open('$OUT/line.json').read());print('[$v] forced collective value', d['value'], d['config']['dist_backend'], d['config']['ids_crc32_rank0'])"
done | tee $OUT/nccl1_variants.txt
python3 bench.py $C 2>/dev/null | grep '^{"metric' | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('replicated value', d['value'], d['config']['ids_crc32_rank0'])" | tee -a $OUT/nccl1_variants.txt
