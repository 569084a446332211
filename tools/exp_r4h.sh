#!/bin/bash
# round-4 session-2 experiment set (GPU box): idle / concurrency of the step, shader clock under the GEMMs, priority and stagger builds
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4h
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
COMMON="--steps 20 --warmup 3 --settle-steps 16 --no-cpu-baseline --no-roofline --no-one-stream --no-h2d --sustained-seconds 0"
rocprofv3 --kernel-trace --output-format csv -d $OUT/two -- python3 $ROOT/bench.py $COMMON > $OUT/two.log 2>&1
python3 $ROOT/tools/trace_idle.py $(find $OUT/two -name "*kernel_trace.csv" | head -1) 0.25 > $OUT/idle_two_streams.txt 2>&1
rm -rf $OUT/two
rocprofv3 --kernel-trace --output-format csv -d $OUT/one -- python3 $ROOT/bench.py $COMMON --no-overlap > $OUT/one.log 2>&1
python3 $ROOT/tools/trace_idle.py $(find $OUT/one -name "*kernel_trace.csv" | head -1) 0.25 > $OUT/idle_one_stream.txt 2>&1
rm -rf $OUT/one
echo idle done
cd $ROOT
MTGV_SP_STAMPS=1 python3 tools/gemm_trace.py $OUT/trace_stamps.csv > $OUT/trace_stamps.txt 2>&1
python3 tools/sp_stamps.py $OUT/trace_stamps.csv.stamps > $OUT/sp_stamps.txt 2>&1
rm -f $OUT/trace_stamps.csv.stamps
echo stamps done
REPS=4 python3 tools/lib_ab.py base prio1 prio2 stg base > $OUT/lib_ab.txt 2>&1
cat $OUT/lib_ab.txt
MTGV_LIB_PATH=$ROOT/mtg-vision_amd/mtgv/libmtgv_stg.so timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_random_shapes.py -m gpu -x -q > $OUT/stg_tests.log 2>&1; echo "stg tests rc=$?"
tail -3 $OUT/stg_tests.log
