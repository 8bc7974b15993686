#!/bin/bash
# round 3, first GPU call: new full-size parity tests, the whole GPU suite, the bench lines
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu ) > gpurun_out/r3a_fullsize.txt 2>&1
( time timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py ) > gpurun_out/r3a_tests.txt 2>&1
( time timeout 600 python bench.py ) > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
( time timeout 300 python bench.py --graph powerlaw --no-side-lines --no-hbm-probe --no-cpu-baseline ) > gpurun_out/r3a_bench_powerlaw.json 2> gpurun_out/r3a_bench_powerlaw.err
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r3a_gpus2.out 2>&1; echo "rc=$?" >> gpurun_out/r3a_gpus2.out
tail -3 gpurun_out/r3a_fullsize.txt gpurun_out/r3a_tests.txt gpurun_out/r3a_gpus2.out
head -c 600 gpurun_out/r3a_bench.json; echo; head -c 400 gpurun_out/r3a_bench_powerlaw.json
