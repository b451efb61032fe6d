#!/bin/bash
# MFMA utilisation and LDS behaviour of the prefill GEMMs from PMC counters (one counter group per pass, kernel-trace only)
TAG=${1:-r1}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcg_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
PMCG=("SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU")
for grp in "${PMCG[@]}"; do
  name=$(echo $grp | tr ' ' '+')
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$name -o c -- python3 $GRAFT_REPO_ROOT/tools/pmc_gemm.py > $OUT/$name.log 2>&1
done
cd - > /dev/null
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gemm" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} {sum(v[2:])/max(1,len(v[2:])):16.0f}  (n={len(v)})")
PY
