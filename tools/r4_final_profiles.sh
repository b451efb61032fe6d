# Round-4 closing measurements on one box (run from the repo root on the GPU box):
#   bench.py (default flags) -> gpurun_out/r4_bench_default.json; the same under rocprofv3 --kernel-trace --stats (no CPU leg, no
#   `configs`: the profiler serialises nothing but would multiply the wall time) -> prof_r4_kernel_stats.csv; the decode step's
#   in-graph per-kernel trace; HBM traffic of the dominant GEMV and of the decode attention from separate --pmc passes.
set -x
bash tools/pmc_round4.sh > gpurun_out/r4_pmc.log 2>&1 || { echo "PMC passes failed: profiles/ not updated"; exit 1; }
cp gpurun_out/pmc_r4_hbm_traffic.json profiles/r04_pmc_hbm_traffic.json   # bench.py reads it (same-source hash) for roofline.traffic
python bench.py 2> gpurun_out/r4_bench_default.err | tail -1 > gpurun_out/r4_bench_default.json
bash tools/prof.sh r4 --steps 32 --warmup 8 --no-cpu-baseline --no-configs > gpurun_out/r4_prof.log 2>&1
bash tools/trace_decode.sh r4 2531 1 24 > gpurun_out/r4_trace.log 2>&1
python tools/clip_attn_probe.py > gpurun_out/r4_clip_attn_probe.txt 2>&1

cat gpurun_out/pmc_r4_hbm_traffic.txt; grep "prefill reps" gpurun_out/r4_bench_default.err; tail -12 gpurun_out/r4_trace.log
