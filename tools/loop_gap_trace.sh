#!/bin/bash
# GPU-side gaps between consecutive decode-step replays: bare replays vs _generate's loop (tools/loop_overhead_probe.py under the kernel trace).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/loopgap
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/tools/loop_overhead_probe.py > $OUT/run.log 2>&1
cd - > /dev/null
tail -3 $OUT/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
ends = [(s, e) for s, e, n in rows if n.startswith("k_step_end")]
begins = [(s, e) for s, e, n in rows if n.startswith("k_step_begin")]
print("steps traced:", len(begins), len(ends))
gaps = []
bi = 0
for (s, e) in ends:
    while bi < len(begins) and begins[bi][0] < e: bi += 1
    if bi < len(begins): gaps.append((begins[bi][0] - e) / 1e3)
# phases: 8 warm-up + 128 bare replays, then 2 x 128 loop steps
def stat(name, g):
    g = sorted(g)
    print(f"{name}: n {len(g)}  mean {sum(g) / len(g):.2f} us  median {g[len(g) // 2]:.2f}  p90 {g[int(len(g) * .9)]:.2f}  max {g[-1]:.2f}")
stat("bare replays   (gap step_end -> next step_begin)", gaps[8:8 + 127])
stat("generate loop 1", gaps[8 + 128 + 1:8 + 128 + 127])
stat("generate loop 2", gaps[8 + 256 + 1:8 + 256 + 127])
PY
rm -f $(find $OUT -name '*kernel_trace.csv')
