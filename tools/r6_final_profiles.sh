#!/bin/bash
# Round-6 closing measurements on one box (run from the repo root on the GPU box through gpurun):
#   PMC passes (HBM traffic of the gate_up GEMV and of the SHIPPED fused attention kernel; MFMA counters of the prefill's dominant GEMM),
#   bench.py with default flags, the same under rocprofv3 --kernel-trace --stats (no CPU leg, no `configs`), the decode step's in-graph
#   per-kernel trace for config 2 and config 5 (projections labelled by their place in the step), config 5's floor table, and the kernel
#   timeline of one prefill (config 2 and config 1).  Everything lands in gpurun_out/; the summaries are copied to profiles/r06_* by hand.
set -x
# (the PMC passes are tools/pmc_round6.sh: run separately, they can hang in the profiler start-up on this pool)

python bench.py 2> gpurun_out/r6_bench_default.err | tail -1 > gpurun_out/r6_bench_default.json
bash tools/prof.sh r6 --steps 32 --warmup 8 --no-cpu-baseline --no-configs > gpurun_out/r6_prof.log 2>&1
bash tools/trace_decode.sh r6 2531 1 24 > gpurun_out/r6_trace.log 2>&1
P3V_FP8=1 P3V_QCACHE=1 bash tools/trace_decode.sh r6c5 2531 1 24 > gpurun_out/r6c5_trace.log 2>&1
python tools/c5_floor.py gpurun_out/trace_r6c5_summary.txt > gpurun_out/c5_floor_r6.txt 2>&1
bash tools/prefill_trace.sh r6 > gpurun_out/r6_prefill_trace.log 2>&1
bash tools/prefill_trace.sh r6c1 c1 > gpurun_out/r6c1_prefill_trace.log 2>&1
grep "prefill reps" gpurun_out/r6_bench_default.err; tail -9 gpurun_out/r6_trace.log; cut -c1-1500 gpurun_out/r6_bench_default.json
