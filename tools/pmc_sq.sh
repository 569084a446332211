#!/bin/bash
# SQ counters per kernel over one pipeline step (GPU box): tools/pmc_sq.sh [tag]
# Separate --pmc passes (kernel trace only, no other trace domains); tools/pmc_sq.py prints per-kernel ratios.
set -e
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" \
         "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/gemm_trace.py $OUT/trace_p$i.csv > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; echo "pass $i failed"; }
done
python3 $ROOT/tools/pmc_sq.py $OUT > $OUT/summary.txt
tail -40 $OUT/summary.txt
