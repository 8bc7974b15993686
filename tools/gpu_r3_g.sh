#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or linear" ) > gpurun_out/r3g_tests.txt 2>&1
timeout 300 python tools/gemm_bench.py > gpurun_out/r3g_gemm_bench.jsonl 2> gpurun_out/r3g_gemm_bench.err
timeout 300 python tools/cfgs_profile.py 30 > gpurun_out/r3g_cfgs_profile.txt 2>&1
tail -n 5 gpurun_out/r3g_tests.txt; grep -E '"N": (150000|157000|300000|40000)' gpurun_out/r3g_gemm_bench.jsonl; tail -2 gpurun_out/r3g_gemm_bench.err; head -4 gpurun_out/r3g_cfgs_profile.txt
