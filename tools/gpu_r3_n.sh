#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or linear or spmm" ) > gpurun_out/r3n_tests.txt 2>&1
( time PYGDA_AMD_SPMM_LDS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spmm or kstep or prop_gcn" ) > gpurun_out/r3n_tests_lds.txt 2>&1
timeout 300 python tools/spmm_sweep.py --big > gpurun_out/r3n_sweep_shuffle.jsonl 2> gpurun_out/r3n_sweep.err
PYGDA_AMD_SPMM_LDS=1 timeout 300 python tools/spmm_sweep.py --big > gpurun_out/r3n_sweep_lds.jsonl 2>> gpurun_out/r3n_sweep.err
timeout 300 python tools/cfgs_profile.py 40 > gpurun_out/r3n_cfgs_profile.txt 2>&1
tail -n 4 gpurun_out/r3n_tests.txt; tail -n 4 gpurun_out/r3n_tests_lds.txt
python - <<'P'
import json
a=[json.loads(l) for l in open("gpurun_out/r3n_sweep_shuffle.jsonl") if l.startswith("{")]
b=[json.loads(l) for l in open("gpurun_out/r3n_sweep_lds.jsonl") if l.startswith("{")]
for x,y in zip(a,b): print(x["case"], x["d"], "shuffle us", x["us"], "lds us", y["us"])
P
head -4 gpurun_out/r3n_cfgs_profile.txt
