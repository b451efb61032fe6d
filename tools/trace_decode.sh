#!/bin/bash
# usage: tools/trace_decode.sh <tag> [ctx] [B] [steps]  -> per-kernel in-graph durations and gaps of the decode step
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/tools/decode_replay.py "$@" > $OUT/run.log 2>&1
cd - > /dev/null
python3 - "$OUT" "$TAG" <<'PY'
import csv, sys, glob, collections
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# decode section = from the first step's k_step_begin on (batches of 9+ rows run k_gemm_skinny inside the step: the prefill's GEMM
# kernels are no marker for where it ends)
import re
# the step's two ends: the launches p3v_step_begin / p3v_step_end, or -- round 6, B = 1 on bf16 weights -- the first / last projection
# that carries them (k_gemv3_step<.., 1> / <.., 2>, and their _f8 / _q4 twins)
def is_begin(n): return "k_step_begin" in n or re.search(r"k_gemv3(_f8|_q4)?_step<[^>]*, 1>", n) is not None
def is_end(n): return "k_step_end" in n or "k_store_token" in n or re.search(r"k_gemv3(_f8|_q4)?_step<[^>]*, 2>", n) is not None
first_step = min(i for i, r in enumerate(rows) if is_begin(r["Kernel_Name"]))
dec = rows[first_step:]
# drop the first step(s): eager warm-up + first replay
names = [r["Kernel_Name"] for r in dec]
idx = [j for j, m in enumerate(names) if is_end(m)]
steps = [(idx[k] + 1, idx[k + 1] + 1) for k in range(len(idx) - 1)]
steps = steps[4:]          # skip warm-up + early replays
agg = collections.OrderedDict(); gaps = []
tot = 0
for a, b in steps:
    seg = dec[a:b]
    # Which projection a GEMV launch is cannot be read off its name (qkv, gate_up and lm_head share an instantiation and a grid): it
    # is labelled by its PLACE in the step -- per layer the order is qkv, attention (+ o_proj), [o_proj], gate_up, down; the last
    # projection of the step is the lm_head (the roofline rows must be separable from this file alone).
    proj = [k for k, r in enumerate(seg) if "k_gemv" in r["Kernel_Name"]]
    attn = [k for k, r in enumerate(seg) if "k_attn_decode" in r["Kernel_Name"]]
    per_layer = max(1, (len(proj) - 1) // max(1, len(attn)))
    names = ["qkv", "gate_up", "down"] if per_layer == 3 else ["qkv", "o_proj", "gate_up", "down"] if per_layer == 4 else None
    role = {}
    if names:
        for n_, k in enumerate(proj[:-1]):
            role[k] = names[n_ % per_layer]
        role[proj[-1]] = "lm_head"
    for k, r in enumerate(seg):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        nm = r["Kernel_Name"].replace("void ", "").split("(")[0][:40]
        if k in role: nm = f"{nm} [{role[k]}]"
        key = (nm, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
        agg.setdefault(key, []).append(d)
        if k > 0: gaps.append((int(r["Start_Timestamp"]) - int(seg[k - 1]["End_Timestamp"])) / 1e3)
    tot += (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
n = len(steps)
lines = [f"decode step (in-graph, rocprofv3 kernel trace, {n} steps averaged): {tot/n:.1f} us/step, "
         f"{len(dec[steps[0][0]:steps[0][1]])} kernels/step, mean gap {sum(gaps)/len(gaps):.2f} us, total gaps {sum(gaps)/n:.1f} us/step"]
for (name, grid), v in agg.items():
    lines.append(f"  {name:52s} grid {grid:>8s}  calls/step {len(v)/n:5.1f}  avg {sum(v)/len(v):8.2f} us  total/step {sum(v)/n:8.1f} us")
open(f"{out}/../trace_{tag}_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -f $(find $OUT -name '*kernel_trace.csv')
