#!/bin/bash
# tools/sweep_bench.sh <label> "<ENV=.. ENV=..>" "<ENV=..>" ... : the headline bench under each environment setting, 2 rounds alternating
LABEL=$1; shift
mkdir -p gpurun_out/r6
OUT=gpurun_out/r6/sweep_${LABEL}.txt
: > $OUT
for i in 1 2; do
  for cfg in "$@"; do
    env $cfg python bench.py --no-cpu-baseline --no-configs --steps 128 --warmup 8 $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s value %.2f tok/s %.4f ms | device %.2f tok/s %.4f ms' % ('$cfg', d['value'], d['ms_per_step'], d['device_rate']['tokens_per_s'], d['device_rate']['ms_per_step']))" >> $OUT
  done
done
cat $OUT
