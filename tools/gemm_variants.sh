#!/bin/bash
# Compare the row split chosen by p3v_gemm's round-packing model with all-small and all-big tiles (run on the GPU box).
echo "== all 128x128";  P3V_GEMM_BIG_ROWS=0 python tools/bench_kernels.py gemm
echo "== all 256x256";  P3V_GEMM_BIG_ROWS=1000000 python tools/bench_kernels.py gemm
echo "== model";        python tools/bench_kernels.py gemm
