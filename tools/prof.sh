#!/bin/bash
# usage: tools_prof.sh <tag> <bench args...>   (run on the GPU box from the repo root)
# rocprofv3 kernel-trace + stats of bench.py; summary copied to gpurun_out/prof_<tag>_kernel_stats.csv
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
cd - > /dev/null
find $OUT -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $PWD/gpurun_out/prof_${TAG}_kernel_stats.csv
rm -f $(find $OUT -name '*kernel_trace.csv')  # large
tail -3 $OUT/bench.err; cat $OUT/bench.json
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$PWD/gpurun_out/prof_${TAG}_kernel_stats.csv")))
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'pct':>6s}")
for r in rows[:25]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:9.2f} {r['Percentage']:>6s}")
PY
