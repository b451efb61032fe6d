// Host-side runtime helpers behind the C ABI: device properties, hipGraph
// capture/replay of a launch sequence (the decode step is replayed as one
// graph: ~170 short weight-streaming kernels per token would otherwise be
// host-launch-bound), and HIP-event timing on the launch stream.
#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "p3v_common.h"

extern "C" int p3v_version(void) { return P3V_VERSION; }

extern "C" const char* p3v_strerror(int code) {
  switch (code) {
    case P3V_OK: return "ok";
    case P3V_ERR_ARG: return "invalid argument (shape, alignment or null pointer)";
    case P3V_ERR_LAUNCH: return "kernel launch failed";
    case P3V_ERR_UNSUPPORTED: return "unsupported combination";
    case P3V_ERR_HIP: return "HIP runtime call failed";
    default: return "unknown error";
  }
}

// ---- launch-policy knobs (p3v_common.h): one table, read from the environment once
namespace {
struct Knob { const char* name; int P3vTuning::*field; int dflt; };
const Knob kKnobs[] = {
    {"gemm_big_rows", &P3vTuning::gemm_big_rows, -1},       {"gemm_no_splitk", &P3vTuning::gemm_no_splitk, 0},
    {"gemm_splitk_max_m", &P3vTuning::gemm_splitk_max_m, 1024}, {"gemm_splitk_max_s", &P3vTuning::gemm_splitk_max_s, 8},
    {"gemm_splitk_wgs", &P3vTuning::gemm_splitk_wgs, 256},  {"gemm_128", &P3vTuning::gemm_128, 0},
    {"gemm_persistent", &P3vTuning::gemm_persistent, 1},     {"gemm_no_qkv_fuse", &P3vTuning::gemm_no_qkv_fuse, 0},
    {"gemm_f8_narrow", &P3vTuning::gemm_f8_narrow, -1},     {"attn_no_dma", &P3vTuning::attn_no_dma, 0},
    {"attn_old", &P3vTuning::attn_old, 0},                  {"attn_pp", &P3vTuning::attn_pp, -1},
    {"attn_il", &P3vTuning::attn_il, -1},                   {"attn_il_waves", &P3vTuning::attn_il_waves, -1},
                      {"combine_g", &P3vTuning::combine_g, -1},
    {"kvq_old", &P3vTuning::kvq_old, 0},                    {"q8_old", &P3vTuning::q8_old, 0},
    {"gemv_no_mfma", &P3vTuning::gemv_no_mfma, 0},          {"gemv_no_mfma8", &P3vTuning::gemv_no_mfma8, 0},
    {"gemv_wpc", &P3vTuning::gemv_wpc, 8},                  {"gemv8_wgs", &P3vTuning::gemv8_wgs, 256},
    {"gemv_variant", &P3vTuning::gemv_variant, 3},          {"gemv_rows", &P3vTuning::gemv_rows, 1},
    {"gemv8_min", &P3vTuning::gemv8_min, 2},                {"gemv_mfma8", &P3vTuning::gemv_mfma8, 1},
    {"gemv_f8_wpc", &P3vTuning::gemv_f8_wpc, 16},           {"gemv_q4_wpc", &P3vTuning::gemv_q4_wpc, 8},
    {"gemv_wpw", &P3vTuning::gemv_wpw, 0},                  {"gemv_q4_rows_wgs", &P3vTuning::gemv_q4_rows_wgs, 256},
    {"gemv_q4_rows8", &P3vTuning::gemv_q4_rows8, 1},
    {"gemm_no_skinny", &P3vTuning::gemm_no_skinny, 0},      {"gemm_skinny_max_m", &P3vTuning::gemm_skinny_max_m, 256},
    {"gemm_skinny_s", &P3vTuning::gemm_skinny_s, 0},            {"gemm_skinny_tm128", &P3vTuning::gemm_skinny_tm128, 0},
    {"gemm_rows", &P3vTuning::gemm_rows, 1},
    {"attn_fo_map", &P3vTuning::attn_fo_map, 2},              {"attn_fo_map_q8", &P3vTuning::attn_fo_map_q8, 0},
};
P3vTuning g_tuning;
std::once_flag g_tuning_once;
void tuning_init() {
  for (const Knob& k : kKnobs) {
    char env[64] = "P3V_";
    size_t n = 4;
    for (const char* c = k.name; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)toupper((unsigned char)*c);
    env[n] = 0;
    const char* e = getenv(env);
    // presence-only switches (P3V_GEMM_128=, P3V_ATTN_OLD=...) historically counted as "on" whatever their value
    g_tuning.*(k.field) = !e ? k.dflt : !strcmp(e, "auto") ? -1 : (*e == 0 ? 1 : atoi(e));
  }
}
}  // namespace

const P3vTuning& p3v_tuning() {
  std::call_once(g_tuning_once, tuning_init);
  return g_tuning;
}

extern "C" int p3v_set_tuning(const char* name, int value) {
  if (!name) return P3V_ERR_ARG;
  std::call_once(g_tuning_once, tuning_init);
  for (const Knob& k : kKnobs)
    if (!strcmp(k.name, name)) { g_tuning.*(k.field) = value; return P3V_OK; }
  return P3V_ERR_ARG;
}

extern "C" int p3v_get_tuning(const char* name, int* value) {
  if (!name || !value) return P3V_ERR_ARG;
  std::call_once(g_tuning_once, tuning_init);
  for (const Knob& k : kKnobs)
    if (!strcmp(k.name, name)) { *value = g_tuning.*(k.field); return P3V_OK; }
  return P3V_ERR_ARG;
}

extern "C" int p3v_device_props(int device, p3v_props_t* out) {
  if (!out) return P3V_ERR_ARG;
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, device) != hipSuccess) return P3V_ERR_HIP;
  memset(out, 0, sizeof(*out));
  out->cu_count = pr.multiProcessorCount;
  out->lds_per_cu = (int)pr.maxSharedMemoryPerMultiProcessor;
  out->wave_size = pr.warpSize;
  out->clock_khz = pr.clockRate;
  out->mem_clock_khz = pr.memoryClockRate;
  out->mem_bus_bits = pr.memoryBusWidth;
  out->hbm_bytes = (int64_t)pr.totalGlobalMem;
  strncpy(out->arch, pr.gcnArchName, sizeof(out->arch) - 1);
  return P3V_OK;
}

extern "C" int p3v_graph_begin(void* stream) {
  return hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) == hipSuccess ? P3V_OK : P3V_ERR_HIP;
}

extern "C" int p3v_graph_end(void* stream, void** graph_exec_out) {
  if (!graph_exec_out) return P3V_ERR_ARG;
  hipGraph_t g = nullptr;
  if (hipStreamEndCapture((hipStream_t)stream, &g) != hipSuccess || !g) return P3V_ERR_HIP;
  hipGraphExec_t ge = nullptr;
  const hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return P3V_ERR_HIP;
  *graph_exec_out = (void*)ge;
  return P3V_OK;
}

extern "C" int p3v_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) return P3V_ERR_ARG;
  return hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream) == hipSuccess ? P3V_OK : P3V_ERR_HIP;
}

extern "C" int p3v_graph_destroy(void* graph_exec) {
  if (!graph_exec) return P3V_OK;
  return hipGraphExecDestroy((hipGraphExec_t)graph_exec) == hipSuccess ? P3V_OK : P3V_ERR_HIP;
}

extern "C" int p3v_event_create(void** ev) {
  if (!ev) return P3V_ERR_ARG;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return P3V_ERR_HIP;
  *ev = (void*)e;
  return P3V_OK;
}
extern "C" int p3v_event_record(void* ev, void* stream) {
  return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? P3V_OK : P3V_ERR_HIP;
}
extern "C" int p3v_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!ms) return P3V_ERR_ARG;
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return P3V_ERR_HIP;
  return hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? P3V_OK : P3V_ERR_HIP;
}
extern "C" int p3v_event_destroy(void* ev) {
  return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? P3V_OK : P3V_ERR_HIP;
}
