#!/bin/bash
# round 3, second GPU call: hub rows in the one-launch K-step kernel
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kstep" ) > gpurun_out/r3b_kstep_tests.txt 2>&1
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "training_step" ) > gpurun_out/r3b_fullsize.txt 2>&1
( time timeout 300 python bench.py --graph powerlaw --no-side-lines --no-hbm-probe --no-cpu-baseline ) > gpurun_out/r3b_bench_powerlaw.json 2> gpurun_out/r3b_bench_powerlaw.err
( time timeout 300 python bench.py --no-side-lines --no-hbm-probe --no-cpu-baseline ) > gpurun_out/r3b_bench_uniform.json 2> gpurun_out/r3b_bench_uniform.err
tail -n 3 gpurun_out/r3b_kstep_tests.txt gpurun_out/r3b_fullsize.txt
head -c 300 gpurun_out/r3b_bench_powerlaw.json; echo; head -c 300 gpurun_out/r3b_bench_uniform.json
