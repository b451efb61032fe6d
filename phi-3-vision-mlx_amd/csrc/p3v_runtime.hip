// Host-side runtime helpers behind the C ABI: device properties, hipGraph
// capture/replay of a launch sequence (the decode step is replayed as one
// graph: ~170 short weight-streaming kernels per token would otherwise be
// host-launch-bound), and HIP-event timing on the launch stream.
#include <string.h>

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
