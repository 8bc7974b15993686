#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_configs.py -x -q -m gpu ) > gpurun_out/r3i_tests.txt 2>&1
timeout 300 python tools/cfgs_profile.py 30 > gpurun_out/r3i_cfgs_profile.txt 2>&1
PYGDA_AMD_INTERIOR_KSTEP=0 timeout 300 python tools/cfgs_profile.py 30 > gpurun_out/r3i_cfgs_profile_full.txt 2>&1
timeout 600 python tools/spmm_slab_probe.py > gpurun_out/r3i_slab_probe.jsonl 2> gpurun_out/r3i_slab_probe.err
tail -n 5 gpurun_out/r3i_tests.txt; head -4 gpurun_out/r3i_cfgs_profile.txt; head -4 gpurun_out/r3i_cfgs_profile_full.txt; tail -3 gpurun_out/r3i_slab_probe.jsonl; tail -2 gpurun_out/r3i_slab_probe.err
