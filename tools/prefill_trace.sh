#!/bin/bash
# usage: tools/prefill_trace.sh <tag> [c1]  -> kernel timeline (durations + gaps) of ONE prefill of bench.py's request, from rocprofv3 --kernel-trace
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ptrace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/tools/prefill_once.py 4 "$@" > $OUT/run.log 2>&1
cd - > /dev/null
cat $OUT/run.log | tail -5
python3 - "$OUT" "$TAG" <<'PY'
import csv, sys, glob, collections
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "spin_kernel" in r["Kernel_Name"]]
seg = rows[marks[-1] + 1:]
t0 = int(rows[marks[-1]]["End_Timestamp"])
agg = collections.OrderedDict()
busy = 0; gaps = 0; prev_end = t0; big_gaps = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
    busy += (e - s) / 1e3
    g = (s - prev_end) / 1e3
    if g > 0: gaps += g
    if g > 8: big_gaps.append((round((s - t0) / 1e3), round(g, 1), name))
    prev_end = max(prev_end, e)
wall = (prev_end - t0) / 1e3
lines = [f"one prefill (rocprofv3 kernel trace): first marker -> last kernel end {wall:.0f} us; kernels busy {busy:.0f} us; gaps {gaps:.0f} us in {len(seg)} launches"]
lines.append(f"{'kernel':62s} {'grid':>8s} {'calls':>6s} {'total us':>10s} {'avg us':>8s}")
for (n, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"{n:62s} {g:>8s} {c:6d} {t:10.1f} {t / c:8.2f}")
lines.append("gaps > 8 us (at us, gap us, next kernel): " + str(big_gaps[:40]))
open(f"gpurun_out/prefill_trace_{tag}.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/*/  # traces are large
