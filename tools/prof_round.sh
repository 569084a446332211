#!/bin/bash
# Round profile set (GPU box), one call:  tools/prof_round.sh r04   -> gpurun_out/prof_r04/
#   kernel statistics of the default bench (two streams) and of the single-stream bench, of the one-rank forced-collective
#   (RCCL) bench, the HBM traffic counter passes, the SQ counter passes, the per-stage probe.
# Every rocprofv3 line starts the program itself after "--" (no env / shell hop: the profiler has the GPU initialised).
set -e
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
COMMON="--steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-one-stream --no-h2d --sustained-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/two_streams -- python3 $ROOT/bench.py $COMMON > $OUT/two_streams.log 2>&1
# (every kernel alone on the GPU: one pipeline stream AND the detector's internal fork-join off)
MTGV_DET_FORK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/no_overlap -- python3 $ROOT/bench.py $COMMON --no-overlap > $OUT/no_overlap.log 2>&1
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 MTGV_FORCE_COLLECTIVE=1 \
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nccl1 -- python3 $ROOT/bench.py --gpus 1 $COMMON --settle-steps 16 > $OUT/nccl1.log 2>&1
for d in two_streams no_overlap nccl1; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/${TAG}_f16x3_kernel_stats_bench_$d.csv
  grep "^{\"metric\"" $OUT/$d.log > $OUT/${TAG}_bench_$d.json
  rm -rf $OUT/$d
done
cd $ROOT
tools/pmc_traffic.sh f16x3 > $OUT/pmc_traffic.log 2>&1
cp $ROOT/gpurun_out/pmc_traffic_f16x3/traffic.json $OUT/${TAG}_traffic_f16x3.json
rm -rf $ROOT/gpurun_out/pmc_traffic_f16x3/FETCH_SIZE $ROOT/gpurun_out/pmc_traffic_f16x3/WRITE_SIZE
tools/pmc_sq.sh $TAG > $OUT/pmc_sq.log 2>&1 || true
cp $ROOT/gpurun_out/pmc_$TAG/summary.txt $OUT/${TAG}_sq_counters_step.txt || true
rm -rf $ROOT/gpurun_out/pmc_$TAG/p1 $ROOT/gpurun_out/pmc_$TAG/p2 $ROOT/gpurun_out/pmc_$TAG/p3 $ROOT/gpurun_out/pmc_$TAG/p4
# idle time / kernel concurrency of the timed region from plain kernel traces (the CSVs are large: analysed here, not kept)
cd /tmp
for m in two_streams no_overlap; do
  X=""; [ $m = no_overlap ] && X="--no-overlap"
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$m -- python3 $ROOT/bench.py --steps 20 --warmup 3 --settle-steps 16 --no-cpu-baseline --no-roofline --no-one-stream --no-h2d --sustained-seconds 0 $X > $OUT/kt_$m.log 2>&1
  python3 $ROOT/tools/trace_idle.py $(find $OUT/kt_$m -name "*kernel_trace.csv" | head -1) 0.10 > $OUT/${TAG}_idle_$m.txt 2>&1 || true
  rm -rf $OUT/kt_$m
done
cd $ROOT
python3 tools/perf_probe.py > $OUT/${TAG}_perf_probe.txt 2>&1
python3 tools/det_probe.py > $OUT/${TAG}_det_probe.txt 2>&1
head -12 $OUT/${TAG}_f16x3_kernel_stats_bench_no_overlap.csv | cut -c1-160
