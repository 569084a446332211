#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $ROOT
COMMON="--no-cpu-baseline --no-roofline --no-h2d --sustained-seconds 0"
for v in "" "MTGV_MATCH_STREAM=0" "MTGV_STREAM_PRIO=none" "MTGV_CROP_STAGE=enc" "MTGV_MATCH_STREAM=0 MTGV_STREAM_PRIO=none MTGV_CROP_STAGE=enc"; do
  r=$(env $v python3 bench.py $COMMON 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['config']['one_stream_value'], d['config']['unsettled_value'])")
  echo "[$v] value one_stream unsettled: $r"
done
