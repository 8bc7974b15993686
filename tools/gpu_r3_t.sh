#!/bin/bash
# functional check of the N-rank bench path on the 1-GPU box (ranks share the GPU, gloo)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "two_rank" ) > gpurun_out/r3t_tests.txt 2>&1
tail -5 gpurun_out/r3t_tests.txt
( time timeout 600 python bench.py --gpus 2 --share-gpus --steps 6 --warmup 2 ) > gpurun_out/r3t_n2.json 2> gpurun_out/r3t_n2.err
echo "rc=$?"; tail -c 300 gpurun_out/r3t_n2.err
python - <<'P'
import json
for n in (2,):
    try:
        d=json.loads(open(f"gpurun_out/r3t_n{n}.json").read().strip().splitlines()[-1])
        print(n, d["n_gpus"], d["ms_per_step"], d["value"], str(d.get("cfgA_replicas"))[:600])
    except Exception as e: print(n, "FAILED", e)
P
