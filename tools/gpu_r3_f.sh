#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sampler.py -x -q -m gpu -k "gemm or linear or sampler or device or loader" ) > gpurun_out/r3f_tests.txt 2>&1
timeout 300 python tools/gemm_bench.py > gpurun_out/r3f_gemm_bench.jsonl 2> gpurun_out/r3f_gemm_bench.err
timeout 600 python tools/spmm_slab_probe.py > gpurun_out/r3f_slab_probe.jsonl 2> gpurun_out/r3f_slab_probe.err
timeout 300 python tools/cfgs_profile.py 30 > gpurun_out/r3f_cfgs_profile.txt 2>&1
tail -n 5 gpurun_out/r3f_tests.txt; cat gpurun_out/r3f_gemm_bench.jsonl; tail -2 gpurun_out/r3f_gemm_bench.err; cat gpurun_out/r3f_slab_probe.jsonl; tail -3 gpurun_out/r3f_slab_probe.err; head -4 gpurun_out/r3f_cfgs_profile.txt
