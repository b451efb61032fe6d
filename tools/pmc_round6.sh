#!/bin/bash
# Round-6 PMC passes (separate passes, kernel-trace only, per MI355X_MICROARCH.md "rocprofv3 PMC slots"):
#   * HBM traffic of the dominant decode GEMV (gate_up) and of the SHIPPED decode attention (k_attn_decode128_o: attention + merge +
#     o_proj + residual) -> gpurun_out/pmc_r6_hbm_traffic.{json,txt}   (json: with the kernel-source hash bench.py checks)
#   * matrix-pipe busy of the prefill's dominant GEMM (k_gemm256<SILU>): SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES, SQ_WAVE_CYCLES,
#     wait counters -> gpurun_out/pmc_r6_gemm_mfma.txt
# Fails loudly (no output files) when a pass fails or counts too few launches.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the repo copy)}"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_r6
rm -rf $OUT; mkdir -p $OUT
cd /tmp
pass() { # <dir> <counters> <script args...>
  local d=$1 c=$2; shift 2
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$d -o p -- python3 $GRAFT_REPO_ROOT/tools/"$@" > $OUT/$d.log 2>&1
}
if [ -z "${ONLY_STEP:-}" ]; then
pass gemv_fetch FETCH_SIZE pmc_kernel.py
pass gemv_write WRITE_SIZE pmc_kernel.py
pass attn_fetch FETCH_SIZE pmc_round6.py attn_o
pass attn_write WRITE_SIZE pmc_round6.py attn_o
pass gemm_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES" pmc_round6.py gemm
pass gemm_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16" pmc_round6.py gemm || echo "(second GEMM counter set not available on this rocprofv3: skipped)"
fi
# the WHOLE decode step (prefill + 8 graph-replayed greedy steps at the headline context) under P3V_PROFILING=1: no launch of the step waits
# for another workgroup of its own grid (separate o_proj and merge launches), so the pass finishes
export P3V_PROFILING=1
# (the pass takes ~10 s when it runs; one attempt in two hung before its first kernel on this pool -- no output at all, inside the profiler's
#  start-up, not in a kernel -- so: a short bound and up to four attempts)
for attempt in 1 2 3 4; do
  rm -rf $OUT/step_fetch
  if timeout 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/step_fetch -o p -- python3 $GRAFT_REPO_ROOT/tools/decode_replay.py 2531 1 8 > $OUT/step_fetch.log 2>&1; then
    echo "whole-step pass: attempt $attempt finished"; break
  fi
  echo "whole-step pass: attempt $attempt did not finish in 90 s"
done
unset P3V_PROFILING
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
res = {"kernel_source_sha16": h.hexdigest()[:16], "collected_by": "tools/pmc_round6.sh (rocprofv3 --kernel-trace --pmc, one counter per pass)"}
lines = []
# algorithmic bytes: gate_up GEMV = 2 x 8192 x 3072 bf16; attention + o_proj = K and V^T of 2541 keys x 32 heads x 96 (bf16) + W_o 3072 x 3072 bf16
for tag, key, alg in (("gemv", "k_gemv3", 2 * 8192 * 3072 * 2), ("attn", "k_attn_decode128_o", 2 * 2541 * 32 * 96 * 2 + 3072 * 3072 * 2)):
    f = [v for k, d in counters(tag + "_fetch").items() if key in k for v in d.get("FETCH_SIZE", [])]
    w = [v for k, d in counters(tag + "_write").items() if key in k for v in d.get("WRITE_SIZE", [])]
    if os.environ.get("ONLY_STEP"):
        break
    if len(f) <= 4 or len(w) <= 4 or sum(f[4:]) == 0:
        sys.exit(f"{tag}: {len(f)} FETCH_SIZE / {len(w)} WRITE_SIZE launches counted -- a pass failed; nothing written")
    fk = sum(f[4:]) / len(f[4:]); wk = sum(w[4:]) / len(w[4:])
    res[tag] = {"kernel": key, "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk, "algorithmic_bytes_per_launch": alg,
                "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024), "launches": len(f),
                "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported"}
    lines.append(f"{key}: FETCH_SIZE {fk:.0f} KiB x2 + WRITE_SIZE {wk:.0f} KiB = {(2*fk+wk)*1024/1e6:.2f} MB per launch; algorithmic {alg/1e6:.2f} MB -> ratio {(2*fk+wk)*1024/alg:.3f}")
# ---- the whole step: FETCH_SIZE per kernel and launch, decode launches only (grid sizes of the B = 1 step), x2, summed per step
st = collections.defaultdict(list)
for f in glob.glob(f"{out}/step_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            st[(r["Kernel_Name"].split("(")[0], r.get("Grid_Size", "?"))].append(float(r["Counter_Value"]))
steps = 8
n_exec = steps + 1                                # the step runs once eagerly before its capture (model._build_decode_graph), then 8 replays
tot = 0.0
sl = [f"whole decode step under P3V_PROFILING=1 (tools/decode_replay.py 2531 1 {steps}: blind model, 2531-token prompt, one eager + {steps} graph-replayed greedy steps), rocprofv3 --pmc FETCH_SIZE;",
      "per decode kernel and grid: launches per step, mean FETCH_SIZE x2 (gfx950 correction) per launch"]
import re
for (k, g), v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
    if not re.search(r"k_gemv3|k_gemv_|k_attn_decode|k_attn_combine|k_step_|k_argmax", k): continue
    # a kernel + grid that the prefill uses too (the last-row lm_head) has launches beyond n_exec x k: the step's share is the floor
    per_step = len(v) // n_exec
    if per_step == 0: continue
    vv = v[len(v) - per_step * n_exec:]
    mean = sum(vv) / len(vv)
    tot += per_step * mean * 2 * 1024
    sl.append(f"  {k[:44]:44s} grid {g:>8s}  {per_step:3d} per step ({len(v)} launches)  {mean * 2 * 1024 / 1e6:9.3f} MB per launch")
sl.append(f"  sum over the step's launches: {tot / 1e9:.3f} GB per token (algorithmic: 7.445 GB of weights + 1.0 GB of K / V at 2531-2539 keys = 8.44 GB)")
if tot == 0: sys.exit("whole-step pass: no launches counted")
open(f"{out}/../pmc_r6_decode_step_fetch.txt", "w").write("\n".join(sl) + "\n")
print("\n".join(sl))
if not os.environ.get("ONLY_STEP"):
    json.dump(res, open(f"{out}/../pmc_r6_hbm_traffic.json", "w"), indent=1)
    open(f"{out}/../pmc_r6_hbm_traffic.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
if os.environ.get("ONLY_STEP"): sys.exit(0)
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
open(f"{out}/../pmc_r6_gemm_mfma.txt", "w").write("\n".join(gl) + "\n")
print("\n".join(gl))
PY
mkdir -p $OUT/../pmc_r6_step_raw && cp $OUT/step_fetch/*counter_collection.csv $OUT/../pmc_r6_step_raw/ 2>/dev/null || true
rm -rf $OUT/*/
