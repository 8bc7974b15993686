#!/bin/bash
# round 3, third GPU call: device neighbour sampler
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_sampler.py -x -q -m gpu ) > gpurun_out/r3c_sampler_tests.txt 2>&1
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kstep" ) > gpurun_out/r3c_kstep_tests.txt 2>&1
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "fit" ) > gpurun_out/r3c_fullsize.txt 2>&1
B="python bench.py --workload cfgS --no-cpu-baseline --steps 30 --warmup 5"
( time timeout 400 $B ) > gpurun_out/r3c_cfgS_dev.json 2> gpurun_out/r3c_cfgS_dev.err
( time PYGDA_AMD_DEVICE_SAMPLER=0 timeout 400 $B ) > gpurun_out/r3c_cfgS_host.json 2> gpurun_out/r3c_cfgS_host.err
( time taskset -c 0-3 timeout 400 $B ) > gpurun_out/r3c_cfgS_dev_4cores.json 2> gpurun_out/r3c_cfgS_dev_4cores.err
( time PYGDA_AMD_DEVICE_SAMPLER=0 taskset -c 0-3 timeout 600 $B ) > gpurun_out/r3c_cfgS_host_4cores.json 2> gpurun_out/r3c_cfgS_host_4cores.err
tail -n 4 gpurun_out/r3c_sampler_tests.txt gpurun_out/r3c_kstep_tests.txt gpurun_out/r3c_fullsize.txt
for f in dev host dev_4cores host_4cores; do head -c 250 gpurun_out/r3c_cfgS_$f.json; echo; done
