#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or linear" ) > gpurun_out/r3r_tests.txt 2>&1
timeout 300 python tools/gemm_bench.py > gpurun_out/r3r_gemm_cq16.jsonl 2> gpurun_out/r3r_gemm.err
PYGDA_AMD_TALL_CQ=8 timeout 300 python tools/gemm_bench.py > gpurun_out/r3r_gemm_cq8.jsonl 2>> gpurun_out/r3r_gemm.err
grep -E "passed|failed" gpurun_out/r3r_tests.txt
for f in cq16 cq8; do echo == $f; python - <<P
import json
for l in open("gpurun_out/r3r_gemm_$f.jsonl"):
    d=json.loads(l)
    if d["N"] in (150000,300000): print(d["N"],d["K"],"fwd",d["fwd_ours"],d["fwd_blas"],"dgrad",d["dgrad_ours"],d["dgrad_blas"],"frac",d["fwd_ours_frac"])
P
done
