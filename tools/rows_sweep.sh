#!/bin/bash
# batched decode step time at 512 keys for B in the given list, rows kernel on / off (P3V_GEMM_ROWS) -> gpurun_out/r6/rows_sweep.txt
mkdir -p gpurun_out/r6
OUT=gpurun_out/r6/rows_sweep.txt
: > $OUT
for B in "$@"; do
  for v in 0 1; do
    P3V_GEMM_ROWS=$v python tools/step_time.py 512 $B 2>/dev/null | tail -1 >> $OUT
  done
done
cat $OUT
