#!/bin/bash
# Round-4 PMC passes (separate passes, kernel-trace only, per MI355X_MICROARCH.md):
#   * HBM traffic of the dominant decode kernel (gate_up GEMV) and of the decode attention -> gpurun_out/pmc_r4_hbm_traffic.json
#     (with the kernel-source hash bench.py checks before it quotes `roofline.traffic`)
set -euo pipefail                      # a failed pass must not leave a valid-looking JSON with zero traffic behind
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the repo copy)}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_r4
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/gemv_fetch -o f -- python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py > $OUT/gemv_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/gemv_write -o w -- python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py > $OUT/gemv_write.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/attn_fetch -o f -- python3 $GRAFT_REPO_ROOT/tools/pmc_round2.py attn > $OUT/attn_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/attn_write -o w -- python3 $GRAFT_REPO_ROOT/tools/pmc_round2.py attn > $OUT/attn_write.log 2>&1
cd - > /dev/null
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections, json, hashlib, os
out = sys.argv[1]
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
h = hashlib.sha256()
for f in ("p3v_gemv.hip", "p3v_gemv3_body.h", "p3v_common.h"):
    h.update(open(os.path.join(root, "phi-3-vision-mlx_amd", "csrc", f), "rb").read())
res = {"kernel_source_sha16": h.hexdigest()[:16], "collected_by": "tools/pmc_round4.sh (rocprofv3 --kernel-trace --pmc, one counter per pass)"}
lines = []
for tag, key, alg in (("gemv", "k_gemv3", 2 * 8192 * 3072 * 2), ("attn", "k_attn_decode128", 2 * 2541 * 32 * 96 * 2)):
    f = [v for k, d in counters(tag + "_fetch").items() if key in k for v in d.get("FETCH_SIZE", [])]
    w = [v for k, d in counters(tag + "_write").items() if key in k for v in d.get("WRITE_SIZE", [])]
    if len(f) <= 4 or len(w) <= 4 or sum(f[4:]) == 0:
        sys.exit(f"{tag}: {len(f)} FETCH_SIZE / {len(w)} WRITE_SIZE launches counted -- a pass failed; nothing written")
    fk = sum(f[4:]) / max(1, len(f[4:])); wk = sum(w[4:]) / max(1, len(w[4:]))
    res[tag] = {"kernel": key, "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk, "algorithmic_bytes_per_launch": alg,
                "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024), "launches": len(f),
                "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported"}
    lines.append(f"{key}: FETCH_SIZE {fk:.0f} KiB x2 + WRITE_SIZE {wk:.0f} KiB = {(2*fk+wk)*1024/1e6:.2f} MB per launch; algorithmic {alg/1e6:.2f} MB -> ratio {(2*fk+wk)*1024/alg:.3f}")
json.dump(res, open(f"{out}/../pmc_r4_hbm_traffic.json", "w"), indent=1)
open(f"{out}/../pmc_r4_hbm_traffic.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
