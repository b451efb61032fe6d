#!/bin/bash
# Build libp3v.so (all HIP kernels + the C ABI) for gfx950, in-tree.
set -e
cd "$(dirname "$0")"
OUT=../libp3v.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wl,-z,defs"   # -z defs: a kernel whose host stub hipcc dropped fails the link, not the dlopen
hipcc $FLAGS -shared p3v_elementwise.hip p3v_gemm.hip p3v_gemm256.hip p3v_gemm_skinny.hip p3v_gemm_rows.hip p3v_gemm_fp8.hip p3v_gemv.hip p3v_gemv_fp8.hip p3v_gemv_q4.hip p3v_attention.hip p3v_lora.hip p3v_preprocess.hip p3v_runtime.hip -o $OUT "$@"
echo "built $(realpath $OUT)"
