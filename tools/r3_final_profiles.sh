# Round-3 closing measurements on one box (run from the repo root on the GPU box): kernel table, PMC passes, bench + rocprof stats.
set -x
python tools/bench_kernels.py attn > gpurun_out/r3_attn_kernels_final.txt 2>&1
PMC_KERNELS="1 2" bash tools/pmc_attn_prefill.sh 8192 > gpurun_out/r3_pmc_attn.txt 2>&1
bash tools/pmc_round3.sh > gpurun_out/r3_pmc_round3.log 2>&1
bash tools/prof.sh r3fin --steps 20 --warmup 5 > gpurun_out/r3_prof_fin.log 2>&1
python bench.py 2>gpurun_out/bench_r3_final.err | tail -1 > gpurun_out/bench_r3_final.json
python tools/long_ctx.py 32768 2>&1 | tail -2 > gpurun_out/r3_long_ctx.txt
python tools/long_ctx.py 8192 2>&1 | tail -2 >> gpurun_out/r3_long_ctx.txt
grep "prefill reps" gpurun_out/bench_r3_final.err; cat gpurun_out/r3_long_ctx.txt; cat gpurun_out/pmc_r3_hbm_traffic.txt
grep "^attn" gpurun_out/r3_attn_kernels_final.txt
grep "k_attn_prefill\|k_gemm256<6" gpurun_out/prof_r3fin_kernel_stats.csv | head
