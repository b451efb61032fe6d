#!/bin/bash
# Round-2 PMC passes (separate passes, kernel-trace only, per MI355X_MICROARCH.md): HBM traffic of the gate_up GEMV (again) and of
# the 128-key decode attention; MFMA busy cycles of the fp8 GEMM.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_r2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/gemv_fetch -o f -- python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py > $OUT/gemv_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/gemv_write -o w -- python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py > $OUT/gemv_write.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/attn_fetch -o f -- python3 $GRAFT_REPO_ROOT/tools/pmc_round2.py attn > $OUT/attn_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/attn_write -o w -- python3 $GRAFT_REPO_ROOT/tools/pmc_round2.py attn > $OUT/attn_write.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/f8_mfma -o c -- python3 $GRAFT_REPO_ROOT/tools/pmc_round2.py fp8gemm > $OUT/f8_mfma.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/f8_lds -o c -- python3 $GRAFT_REPO_ROOT/tools/pmc_round2.py fp8gemm > $OUT/f8_lds.log 2>&1
cd - > /dev/null
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
def durations(sub, key):
    d = []
    for f in glob.glob(f"{out}/{sub}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]: d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return d
lines = []
res = {}
for tag, key, alg in (("gemv", "k_gemv3", 2 * 8192 * 3072 * 2), ("attn", "k_attn_decode128", 2 * 2541 * 32 * 96 * 2)):
    f = [v for k, d in counters(tag + "_fetch").items() if key in k for v in d.get("FETCH_SIZE", [])]
    w = [v for k, d in counters(tag + "_write").items() if key in k for v in d.get("WRITE_SIZE", [])]
    fk = sum(f[4:]) / max(1, len(f[4:])); wk = sum(w[4:]) / max(1, len(w[4:]))
    res[tag] = {"kernel": key, "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk, "algorithmic_bytes_per_launch": alg,
                "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024), "launches": len(f),
                "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported"}
    lines.append(f"{key}: FETCH_SIZE {fk:.0f} KiB x2 + WRITE_SIZE {wk:.0f} KiB = {(2*fk+wk)*1024/1e6:.2f} MB per launch; algorithmic {alg/1e6:.2f} MB -> ratio {(2*fk+wk)*1024/alg:.3f}")
acc = counters("f8_mfma"); lds = counters("f8_lds")
for k in acc:
    if "gemm256_f8" not in k: continue
    busy, mf = acc[k].get("SQ_BUSY_CYCLES", []), acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES", [])
    du = durations("f8_mfma", "gemm256_f8")
    lines.append(f"{k}: per-launch SQ_VALU_MFMA_BUSY_CYCLES {mf} SQ_BUSY_CYCLES {busy}")
    lines.append(f"   launch durations under the counters (us): {[round(x, 1) for x in du]}")
    bc, ia = lds.get(k, {}).get("SQ_LDS_BANK_CONFLICT", []), lds.get(k, {}).get("SQ_LDS_IDX_ACTIVE", [])
    if bc and ia: lines.append(f"   LDS bank conflict / active cycles: {sum(bc)/max(1.0,sum(ia)):.4f}")
json.dump(res, open(f"{out}/../pmc_r2_summary.json", "w"), indent=1)
open(f"{out}/../pmc_r2_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
