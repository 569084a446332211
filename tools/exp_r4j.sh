#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4j
mkdir -p $OUT
cd $ROOT
MTGV_ENC_CHUNK_MB=3 timeout -k 10 300 python3 -m pytest tests/test_gpu_encoder.py tests/test_gpu_precision.py -m gpu -x -q > $OUT/chunk_tests.log 2>&1; echo "chunk tests rc=$?"; tail -2 $OUT/chunk_tests.log
MTGV_ENC_CHUNK_MB=3 MTGV_MLP_FUSED=0 timeout -k 10 300 python3 -m pytest tests/test_gpu_encoder.py -m gpu -x -q > $OUT/chunk_tests2.log 2>&1; echo "chunk tests (unfused) rc=$?"; tail -2 $OUT/chunk_tests2.log
{
for rep in 1 2; do
echo "default:";                           python3 tools/perf_probe.py ae_tiny 2>&1 | grep ae_tiny
echo "MLP_FUSED=0:";    MTGV_MLP_FUSED=0   python3 tools/perf_probe.py ae_tiny 2>&1 | grep ae_tiny
for mb in 40 80 120 160; do
echo "CHUNK_MB=$mb:";   MTGV_ENC_CHUNK_MB=$mb python3 tools/perf_probe.py ae_tiny 2>&1 | grep ae_tiny
echo "CHUNK_MB=$mb MLP_FUSED=0:"; MTGV_ENC_CHUNK_MB=$mb MTGV_MLP_FUSED=0 python3 tools/perf_probe.py ae_tiny 2>&1 | grep ae_tiny
done
done
} > $OUT/chunk_probe.txt 2>&1
cat $OUT/chunk_probe.txt
