#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4i
mkdir -p $OUT
cd $ROOT
REPS=4 python3 tools/lib_ab.py base mfma16 base > $OUT/lib_ab_mfma16.txt 2>&1
cat $OUT/lib_ab_mfma16.txt
MTGV_LIB_PATH=$ROOT/mtg-vision_amd/mtgv/libmtgv_mfma16.so MTGV_SP_STAMPS=1 python3 tools/gemm_trace.py $OUT/trace_stamps16.csv > $OUT/trace_stamps16.txt 2>&1
python3 tools/sp_stamps.py $OUT/trace_stamps16.csv.stamps > $OUT/sp_stamps16.txt 2>&1
rm -f $OUT/trace_stamps16.csv.stamps
grep -E "^ *(6[0-9]|7[0-9]|8[0-9]) " $OUT/sp_stamps16.txt | cut -c1-60,120-200
