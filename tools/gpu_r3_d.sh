#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kstep" ) > gpurun_out/r3d_kstep_tests.txt 2>&1
timeout 300 python tools/cfgs_profile.py 30 > gpurun_out/r3d_cfgs_profile.txt 2>&1
PYGDA_AMD_DEVICE_SAMPLER=0 timeout 300 python tools/cfgs_profile.py 30 > gpurun_out/r3d_cfgs_profile_host.txt 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_cfgs -o cfgs -- python $GRAFT_REPO_ROOT/bench.py --workload cfgS --no-cpu-baseline --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/r3d_rocprof_cfgS.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3d_rocprof_cfgS.err
cd $GRAFT_REPO_ROOT
find /tmp/prof_cfgs -name "*kernel_stats*" | head -3
for f in $(find /tmp/prof_cfgs -name "*kernel_stats.csv" | head -1); do head -45 $f > gpurun_out/r3d_cfgS_kernel_stats.csv; done
tail -n 4 gpurun_out/r3d_kstep_tests.txt; head -5 gpurun_out/r3d_cfgs_profile.txt; head -5 gpurun_out/r3d_cfgs_profile_host.txt
