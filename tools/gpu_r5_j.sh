#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "tall_gemm or skinny or gemm" > $O/r5j_tests.txt 2>&1
tail -15 $O/r5j_tests.txt | cut -c1-200
timeout 300 python tools/gemm_bench.py > $O/r5j_gemm_split.jsonl 2> $O/r5j_gemm_split.err
PYGDA_AMD_GEMM_SPLIT_F16=0 timeout 300 python tools/gemm_bench.py > $O/r5j_gemm_fp32.jsonl 2> $O/r5j_gemm_fp32.err
python - <<'PY'
import json
for f in ("r5j_gemm_split", "r5j_gemm_fp32"):
    for l in open(f"gpurun_out/{f}.jsonl"):
        d = json.loads(l)
        if d["N"] >= 150000:
            print(f, d["N"], d["K"], "fwd", d["fwd_ours"], "dgrad", d["dgrad_ours"], "wgrad", d["wgrad_ours"], "blas fwd", d["fwd_blas"])
PY
