#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
st() { grep -E "nr_periods|nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
python -c "import torch; print('torch threads', torch.get_num_threads(), torch.__config__.parallel_info().splitlines()[:8])"
echo "before: $(st)"
for i in 1 2 3; do
python bench.py --workload cfgS --no-cpu-baseline > $O/r5_cg_$i.json 2> $O/r5_cg_$i.err
python - "$i" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5_cg_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 3), d["config"]["host_ms_per_step_max_median"])
PY
echo "after $i: $(st)"
done
OMP_NUM_THREADS=8 python bench.py --workload cfgS --no-cpu-baseline > $O/r5_cg_4.json 2> $O/r5_cg_4.err
echo "after omp8: $(st)"
python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained > $O/r5_cg_5.json 2> $O/r5_cg_5.err
echo "after cfgA: $(st)"
