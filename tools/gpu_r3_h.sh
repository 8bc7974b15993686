#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/pmc_gemm_$tag -- python tools/gemm_pmc.py > /dev/null 2> $O/pmc_gemm_$tag.err
done
python - <<'P'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_gemm_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_tall" in k:
            agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/r3h_gemm_pmc.txt", "w") as out:
    for k, cs in agg.items():
        for c, v in sorted(cs.items()):
            line = f"{k} {c} launches={len(v)} avg={sum(v)/len(v):.4g}"
            print(line); out.write(line + "\n")
P
rm -rf $O/pmc_gemm_*/
