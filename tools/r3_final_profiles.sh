set -x
python tools/bench_kernels.py attn > gpurun_out/r3_attn_kernels.txt 2>&1
build/valu_issue_rate > gpurun_out/r3_valu_issue_rate.txt 2>&1
build/mfma_valu_overlap > gpurun_out/r3_mfma_valu_overlap.txt 2>&1
build/slot_bench > gpurun_out/r3_slot_bench.txt 2>&1
PMC_KERNELS="1 2" bash tools/pmc_attn_prefill.sh 8192 > gpurun_out/r3_pmc_attn.txt 2>&1
bash tools/pmc_round3.sh > gpurun_out/r3_pmc_round3.log 2>&1
tail -3 gpurun_out/r3_pmc_round3.log
cat gpurun_out/r3_attn_kernels.txt
