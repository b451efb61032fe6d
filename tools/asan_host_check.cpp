// Host-side AddressSanitizer check of libp3v.so's LAUNCHER code ("optional -fsanitize=address host build").
// Runs on a box WITHOUT a GPU: every call below returns from the launcher's host-side argument validation or is a pure host
// function (workspace sizes, tuning table, version), so no kernel is launched; ASan watches the launchers' own reads of the
// argument structs, the tuning table's string handling and the static state.  Build + run: tools/asan_host_build.sh
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/p3v.h"

static int fails = 0;
#define EXPECT(cond)                                                      \
  do {                                                                    \
    if (!(cond)) { std::printf("FAIL line %d: %s\n", __LINE__, #cond); ++fails; } \
  } while (0)

int main() {
  EXPECT(p3v_version() > 0);
  // workspace sizing: pure host arithmetic
  EXPECT(p3v_attention_ws_bytes(1, 1, 32, 96, 21) == (int64_t)1 * 32 * 21 * 16 * 98 * 4);
  EXPECT(p3v_attention_ws_bytes(1, 64, 32, 96, 4) == 0);
  EXPECT(p3v_gemm_ws_bytes(128, 3072, 3072, P3V_EPI_NONE) >= 0);
  EXPECT(p3v_gemm_ws_bytes(0, 3072, 3072, P3V_EPI_NONE) == 0);
  // tuning table: names are matched as C strings
  int v = -123;
  EXPECT(p3v_get_tuning("gemm_persistent", &v) == P3V_OK);
  EXPECT(p3v_set_tuning("gemm_persistent", v) == P3V_OK);
  EXPECT(p3v_get_tuning("no_such_knob", &v) != P3V_OK);
  EXPECT(p3v_set_tuning("", 1) != P3V_OK);
  std::vector<char> longname(4096, 'x');
  longname.back() = 0;
  EXPECT(p3v_set_tuning(longname.data(), 1) != P3V_OK);
  // argument validation of the launchers: null pointers, bad sizes, misaligned operands -> error codes, no launch
  EXPECT(p3v_gemm(nullptr, nullptr) == P3V_ERR_ARG);
  p3v_gemm_args_t g;
  std::memset(&g, 0, sizeof g);
  EXPECT(p3v_gemm(&g, nullptr) == P3V_ERR_ARG);
  alignas(16) static uint16_t buf[4096];
  g.A = buf; g.W = buf; g.out = buf; g.M = 16; g.N = 64; g.K = 63; g.lda = 63; g.ldw = 63; g.ldo = 64;   // K % 64 != 0
  EXPECT(p3v_gemm(&g, nullptr) == P3V_ERR_ARG);
  g.K = 64; g.lda = 64; g.ldw = 64; g.A = buf + 1;                                                          // misaligned A
  EXPECT(p3v_gemm(&g, nullptr) == P3V_ERR_ARG);
  g.A = buf; g.epilogue = P3V_EPI_RESID_BF16; g.resid = nullptr;                                            // residual missing
  EXPECT(p3v_gemm(&g, nullptr) == P3V_ERR_ARG);
  g.epilogue = 99;
  EXPECT(p3v_gemm(&g, nullptr) == P3V_ERR_ARG);
  g.epilogue = P3V_EPI_NONE; g.M = 0;                                                                        // empty problem: OK, no launch
  EXPECT(p3v_gemm(&g, nullptr) == P3V_OK);
  EXPECT(p3v_rmsnorm(nullptr, nullptr, nullptr, 1, 3072, 1e-5f, nullptr) == P3V_ERR_ARG);
  EXPECT(p3v_rmsnorm(buf, buf, buf, 0, 3072, 1e-5f, nullptr) == P3V_OK);
  EXPECT(p3v_rmsnorm(buf, buf, buf, 1, 3071, 1e-5f, nullptr) == P3V_ERR_ARG);
  EXPECT(p3v_log_softmax(nullptr, buf, 1, 8, nullptr) == P3V_ERR_ARG);
  EXPECT(p3v_log_softmax(buf, buf, 0, 8, nullptr) == P3V_OK);
  EXPECT(p3v_attention(nullptr, nullptr) == P3V_ERR_ARG);
  p3v_attn_args_t a;
  std::memset(&a, 0, sizeof a);
  EXPECT(p3v_attention(&a, nullptr) == P3V_ERR_ARG);
  p3v_attn_decode_args_t d;
  std::memset(&d, 0, sizeof d);
  EXPECT(p3v_attention_decode(&d, nullptr) != P3V_OK);
  std::printf(fails ? "asan host check: %d FAILED\n" : "asan host check: all launcher argument paths clean (%d failures)\n", fails);
  return fails ? 1 : 0;
}
