#!/bin/bash
# Round-5 PMC passes (separate passes, kernel-trace only, per MI355X_MICROARCH.md "rocprofv3 PMC slots"):
#   * HBM traffic of the dominant decode GEMV (gate_up) and of the SHIPPED decode attention (k_attn_decode128_o: attention + merge +
#     o_proj + residual) -> gpurun_out/pmc_r5_hbm_traffic.{json,txt}   (json: with the kernel-source hash bench.py checks)
#   * matrix-pipe busy of the prefill's dominant GEMM (k_gemm256<SILU>): SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES, SQ_WAVE_CYCLES,
#     wait counters -> gpurun_out/pmc_r5_gemm_mfma.txt
# Fails loudly (no output files) when a pass fails or counts too few launches.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the repo copy)}"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_r5
rm -rf $OUT; mkdir -p $OUT
cd /tmp
pass() { # <dir> <counters> <script args...>
  local d=$1 c=$2; shift 2
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$d -o p -- python3 $GRAFT_REPO_ROOT/tools/"$@" > $OUT/$d.log 2>&1
}
pass gemv_fetch FETCH_SIZE pmc_kernel.py
pass gemv_write WRITE_SIZE pmc_kernel.py
pass attn_fetch FETCH_SIZE pmc_round5.py attn_o
pass attn_write WRITE_SIZE pmc_round5.py attn_o
pass gemm_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES" pmc_round5.py gemm
pass gemm_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16" pmc_round5.py gemm || echo "(second GEMM counter set not available on this rocprofv3: skipped)"
cd - > /dev/null
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections, json, hashlib, os
out = sys.argv[1]
root = os.environ["GRAFT_REPO_ROOT"]
def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
h = hashlib.sha256()
for f in ("p3v_gemv.hip", "p3v_gemv3_body.h", "p3v_common.h"):
    h.update(open(os.path.join(root, "phi-3-vision-mlx_amd", "csrc", f), "rb").read())
res = {"kernel_source_sha16": h.hexdigest()[:16], "collected_by": "tools/pmc_round5.sh (rocprofv3 --kernel-trace --pmc, one counter per pass)"}
lines = []
# algorithmic bytes: gate_up GEMV = 2 x 8192 x 3072 bf16; attention + o_proj = K and V^T of 2541 keys x 32 heads x 96 (bf16) + W_o 3072 x 3072 bf16
for tag, key, alg in (("gemv", "k_gemv3", 2 * 8192 * 3072 * 2), ("attn", "k_attn_decode128_o", 2 * 2541 * 32 * 96 * 2 + 3072 * 3072 * 2)):
    f = [v for k, d in counters(tag + "_fetch").items() if key in k for v in d.get("FETCH_SIZE", [])]
    w = [v for k, d in counters(tag + "_write").items() if key in k for v in d.get("WRITE_SIZE", [])]
    if len(f) <= 4 or len(w) <= 4 or sum(f[4:]) == 0:
        sys.exit(f"{tag}: {len(f)} FETCH_SIZE / {len(w)} WRITE_SIZE launches counted -- a pass failed; nothing written")
    fk = sum(f[4:]) / len(f[4:]); wk = sum(w[4:]) / len(w[4:])
    res[tag] = {"kernel": key, "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk, "algorithmic_bytes_per_launch": alg,
                "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024), "launches": len(f),
                "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported"}
    lines.append(f"{key}: FETCH_SIZE {fk:.0f} KiB x2 + WRITE_SIZE {wk:.0f} KiB = {(2*fk+wk)*1024/1e6:.2f} MB per launch; algorithmic {alg/1e6:.2f} MB -> ratio {(2*fk+wk)*1024/alg:.3f}")
json.dump(res, open(f"{out}/../pmc_r5_hbm_traffic.json", "w"), indent=1)
open(f"{out}/../pmc_r5_hbm_traffic.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
gl = ["prefill GEMM gate_up 2531 x 8192 x 3072 (SiLU epilogue), rocprofv3 --pmc, per launch (mean over launches after the first 2):"]
for sub in ("gemm_mfma", "gemm_wait"):
    for k, d in counters(sub).items():
        if "k_gemm" not in k: continue
        m = {c: sum(v[2:]) / max(1, len(v[2:])) for c, v in d.items()}
        gl.append(f"  {k[:40]:40s} " + "  ".join(f"{c} {x:.4g}" for c, x in sorted(m.items())))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("SQ_BUSY_CU_CYCLES"):
            # SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD-with-MFMA-busy summed over SIMDs (guide: = 32 x N_mfma for 32x32x16; 16 per 16x16x32);
            # SQ_BUSY_CU_CYCLES counts quad-cycles... ratio quoted as reported, with the flop-derived utilisation beside it
            gl.append(f"    MFMA busy / (4 SIMDs x CU busy cycles) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CU_CYCLES']):.3f} (counter units as reported by this rocprofv3; see profiles/README.md)")
open(f"{out}/../pmc_r5_gemm_mfma.txt", "w").write("\n".join(gl) + "\n")
print("\n".join(gl))
PY
rm -rf $OUT/*/
