#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_graph_mode.py tests/test_gpu_parity.py -x -q -m gpu -k "graph_mode or segment_mean or collate or kstep" ) > gpurun_out/r3e_tests.txt 2>&1
timeout 900 bash tools/profile_r3.sh > gpurun_out/r3e_profile.log 2>&1
timeout 600 python tools/spmm_slab_probe.py > gpurun_out/r3e_slab_probe.jsonl 2> gpurun_out/r3e_slab_probe.err
timeout 300 python tools/gemm_bench.py > gpurun_out/r3e_gemm_bench.jsonl 2> gpurun_out/r3e_gemm_bench.err
tail -n 5 gpurun_out/r3e_tests.txt; tail -n 12 gpurun_out/r3e_profile.log; cat gpurun_out/r3e_slab_probe.jsonl; tail -3 gpurun_out/r3e_slab_probe.err; tail -4 gpurun_out/r3e_gemm_bench.jsonl
