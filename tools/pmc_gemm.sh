#!/bin/bash
# usage (GPU box): tools/pmc_gemm.sh <tag> M N K act [res]
set -e
TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py "$@" > $OUT.a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py "$@" > $OUT.b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/c -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py "$@" > $OUT.c.log 2>&1
