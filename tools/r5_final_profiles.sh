#!/bin/bash
# Round-5 closing measurements on one box (run from the repo root on the GPU box through gpurun):
#   PMC passes (HBM traffic of the gate_up GEMV and of the SHIPPED fused attention kernel; MFMA counters of the prefill's dominant GEMM),
#   bench.py with default flags, the same under rocprofv3 --kernel-trace --stats (no CPU leg, no `configs`), the decode step's in-graph
#   per-kernel trace for config 2 and config 5 (projections labelled by their place in the step), config 5's floor table, and the kernel
#   timeline of one prefill (config 2 and config 1).  Everything lands in gpurun_out/; the summaries are copied to profiles/r05_* by hand.
set -x
bash tools/pmc_round5.sh > gpurun_out/r5_pmc.log 2>&1 || { echo "PMC passes failed: profiles/ not updated"; exit 1; }
cp gpurun_out/pmc_r5_hbm_traffic.json profiles/r05_pmc_hbm_traffic.json   # bench.py reads it (same-source hash) for roofline.traffic
python bench.py 2> gpurun_out/r5_bench_default.err | tail -1 > gpurun_out/r5_bench_default.json
bash tools/prof.sh r5 --steps 32 --warmup 8 --no-cpu-baseline --no-configs > gpurun_out/r5_prof.log 2>&1
bash tools/trace_decode.sh r5 2531 1 24 > gpurun_out/r5_trace.log 2>&1
P3V_FP8=1 P3V_QCACHE=1 bash tools/trace_decode.sh r5c5 2531 1 24 > gpurun_out/r5c5_trace.log 2>&1
python tools/c5_floor.py gpurun_out/trace_r5c5_summary.txt > gpurun_out/c5_floor_r5.txt 2>&1
bash tools/prefill_trace.sh r5 > gpurun_out/r5_prefill_trace.log 2>&1
bash tools/prefill_trace.sh r5c1 c1 > gpurun_out/r5c1_prefill_trace.log 2>&1
cat gpurun_out/pmc_r5_hbm_traffic.txt; grep "prefill reps" gpurun_out/r5_bench_default.err; tail -9 gpurun_out/r5_trace.log; cut -c1-1500 gpurun_out/r5_bench_default.json
