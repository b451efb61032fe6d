#!/bin/bash
# HBM traffic of the dominant decode kernel (gate_up GEMV) from PMC counters: separate passes, kernel-trace only,
# per /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports
# HALF the bytes of a wide coalesced read -> x2.
TAG=${1:-r1}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py > $OUT/fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py > $OUT/write.log 2>&1
cd - > /dev/null
python3 - "$OUT" "$TAG" <<'PY'
import csv, sys, glob, collections, json
out, tag = sys.argv[1], sys.argv[2]
res = {}
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(f"{out}/{kind}/**/*counter_collection.csv", recursive=True)
    if not files:
        print("no counter file for", kind, glob.glob(f"{out}/{kind}/**/*", recursive=True)[:10]); continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0])) if r.get("Counter_Name") == ctr and "gemv" in r["Kernel_Name"]]
    res[ctr + "_KiB_per_launch"] = sum(vals[4:]) / max(1, len(vals[4:]))
    res[ctr + "_launches"] = len(vals)
f, w = res.get("FETCH_SIZE_KiB_per_launch", 0.0), res.get("WRITE_SIZE_KiB_per_launch", 0.0)
res["kernel"] = "k_gemv3<1,1,6> gate_up (RMSNorm + SiLU*up), N=8192 K=3072"
res["algorithmic_bytes_per_launch"] = 2 * 8192 * 3072 * 2
res["hbm_bytes_per_launch_corrected"] = int((2 * f + w) * 1024)
res["correction"] = "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported"
print(json.dumps(res, indent=1))
json.dump(res, open(f"{out}/../pmc_{tag}_summary.json", "w"), indent=1)
PY
