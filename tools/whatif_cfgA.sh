#!/bin/bash
# What would the cfg-A step cost WITHOUT a given launch?  (PYGDA_AMD_DBG_SKIP: csrc/gda_common.h -- results are wrong by
# construction, only ms/step is read.)  One line per variant, alternating with the baseline.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B="python bench.py --no-cpu-baseline --no-hbm-probe --no-side-lines --no-sustained"
run() {
  PYGDA_AMD_DBG_SKIP="$2" $B > $O/whatif_$1.json 2> $O/whatif_$1.err
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/whatif_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:14s} {d['ms_per_step']:.4f}   skip={sys.argv[2]}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run base0 ""
run slab k_slab_sum
run transpose k_transpose
run fwdT k_relu_dropout_fwd_T
run bwdT k_relu_dropout_bwd_T
run base1 ""
run wgrad gemm_ex_tn
run wgrad_slab gemm_ex_tn,k_slab_sum
run finalize k_finalize
run tilesplit k_tile_split
run mmd k_mmd_fused
run base2 ""
run cefinal k_ce_final,k_colsum_final,k_stack2
run kstep kstep_colmajor
run allsmall k_slab_sum,k_transpose,k_relu_dropout_fwd_T,k_relu_dropout_bwd_T,k_finalize,k_tile_split,k_ce_final,k_colsum_final,k_stack2
run base3 ""
