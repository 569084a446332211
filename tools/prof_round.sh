#!/bin/bash
# Round profile set (GPU box): kernel statistics of the default bench (two streams) and of the single-stream bench,
# then the HBM traffic counter passes.  tools/prof_round.sh r02   -> gpurun_out/prof_r02/
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/two_streams -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-one-stream > $OUT/two_streams.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/no_overlap -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-overlap > $OUT/no_overlap.log 2>&1
for d in two_streams no_overlap; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/${TAG}_f16x3_kernel_stats_bench_$d.csv
  grep "^{\"metric\"" $OUT/$d.log > $OUT/${TAG}_bench_$d.json
done
$ROOT/tools/pmc_traffic.sh f16x3
cp $ROOT/gpurun_out/pmc_traffic_f16x3/traffic.json $OUT/${TAG}_traffic_f16x3.json
head -12 $OUT/${TAG}_f16x3_kernel_stats_bench_no_overlap.csv
