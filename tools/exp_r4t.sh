#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4t; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for o in matcher-first detector-first; do
  ORDER=$o rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$o -- python3 $ROOT/tools/debug/enqueue_time.py 20 > $OUT/kt_$o.log 2>&1
  echo "== $o"; grep overlapped $OUT/kt_$o.log | tail -1
  python3 $ROOT/tools/debug/queue_map.py $(find $OUT/kt_$o -name "*kernel_trace.csv" | head -1)
  rm -rf $OUT/kt_$o
done > $OUT/queue_map.txt 2>&1
cat $OUT/queue_map.txt
