#!/bin/bash
# One GPU-box call that produces everything profiles/ is built from (run from the repo root):
#   GPU tests, bench in both precisions, rocprofv3 kernel stats (one stream and default two streams), PMC traffic.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/measure
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 || { tail -30 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
timeout -k 10 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout -k 10 300 python bench.py --precision f32 --no-cpu-baseline > $OUT/bench_f32.json 2> $OUT/bench_f32.err
cat $OUT/bench_default.json $OUT/bench_f32.json
cd /tmp; export TMPDIR=/tmp
for MODE in no_overlap default_overlap; do
  FLAG=""; [ $MODE = no_overlap ] && FLAG="--no-overlap"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$MODE -- python3 $ROOT/bench.py --steps 10 --warmup 3 \
    --no-cpu-baseline --no-roofline $FLAG > $OUT/stats_$MODE.log 2>&1
  find $OUT/stats_$MODE -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$MODE.csv \;
  rm -rf $OUT/stats_$MODE
done
cd $ROOT
tools/pmc_traffic.sh f16x3 > $OUT/pmc.log 2>&1
cp $ROOT/gpurun_out/pmc_traffic_f16x3/traffic.json $OUT/traffic_f16x3.json
rm -rf $ROOT/gpurun_out/pmc_traffic_f16x3/FETCH_SIZE $ROOT/gpurun_out/pmc_traffic_f16x3/WRITE_SIZE
head -12 $OUT/kernel_stats_no_overlap.csv
