#!/bin/bash
# A/B of one library knob on the headline bench: tools/ab_bench.sh <ENV_NAME> <value A> <value B> [rounds] [extra bench.py args]
# alternating runs in one box, device rate + reference-defined rate per run -> gpurun_out/r6/ab_<ENV>.txt
ENVN=$1; A=$2; B=$3; R=${4:-2}; shift 4
mkdir -p gpurun_out/r6
OUT=gpurun_out/r6/ab_${ENVN}.txt
: > $OUT
for i in $(seq 1 $R); do
  for v in $A $B; do
    env $ENVN=$v python bench.py --no-cpu-baseline --no-configs --steps 128 --warmup 8 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$ENVN=$v value %.2f tok/s  %.4f ms/step | device %.2f tok/s %.4f ms | prefill %.2f ms' % (d['value'], d['ms_per_step'], d['device_rate']['tokens_per_s'], d['device_rate']['ms_per_step'], d['prefill_ms']))" >> $OUT
  done
done
cat $OUT
