#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $ROOT
C="--no-cpu-baseline --no-roofline --no-h2d --sustained-seconds 0"
for rep in 1 2; do for q in "" 8; do
  if [ -z "$q" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  python3 bench.py $C 2>/dev/null | grep '^{"metric' | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('queues=$q replicated value', d['value'], 'one stream', d['config']['one_stream_value'])"
done; done
