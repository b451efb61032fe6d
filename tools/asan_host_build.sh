#!/bin/bash
# Optional AddressSanitizer HOST build of libp3v.so + the launcher check that runs without a GPU.
# The device code is compiled as usual (GPU ASan / xnack+ objects are not available on this pool: -fno-gpu-sanitize);
# only the host side -- the extern "C" launchers, their argument validation, the tuning table -- is instrumented.
#   tools/asan_host_build.sh            -> build/libp3v_asan.so, build/asan_host_check, runs the check
set -e
cd "$(dirname "$0")/.."
mkdir -p build
SRC=phi-3-vision-mlx_amd/csrc
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan -Wno-unused-result -shared \
  $SRC/p3v_elementwise.hip $SRC/p3v_gemm.hip $SRC/p3v_gemm256.hip $SRC/p3v_gemm_fp8.hip $SRC/p3v_gemv.hip $SRC/p3v_gemv_fp8.hip \
  $SRC/p3v_gemv_q4.hip $SRC/p3v_attention.hip $SRC/p3v_lora.hip $SRC/p3v_preprocess.hip $SRC/p3v_runtime.hip -o build/libp3v_asan.so
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address -shared-libsan tools/asan_host_check.cpp -o build/asan_host_check \
  -Lbuild -lp3v_asan -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 LD_LIBRARY_PATH=build:$(dirname $(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)):/opt/rocm/lib ./build/asan_host_check
