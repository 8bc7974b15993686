#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or linear" ) > gpurun_out/r3l_tests.txt 2>&1
timeout 300 python tools/gemm_bench.py > gpurun_out/r3l_gemm_bench16.jsonl 2> gpurun_out/r3l_gemm_bench.err
PYGDA_AMD_TALL_FWD32=1 timeout 300 python tools/gemm_bench.py > gpurun_out/r3l_gemm_bench32.jsonl 2>> gpurun_out/r3l_gemm_bench.err
tail -n 4 gpurun_out/r3l_tests.txt
for f in 16 32; do echo == $f; grep -E '"N": (150000|300000|40000)' gpurun_out/r3l_gemm_bench$f.jsonl | cut -c1-330; done
