// d2h_engine.hip -- which engine moves a device->host copy, and what it costs a kernel that streams HBM beside it.
//
//   hipcc -O2 --offload-arch=gfx950 -o d2h_engine d2h_engine.hip -lhsa-runtime64 -pthread && ./d2h_engine <mode>
//   modes: hip     hipMemcpyAsync(device -> pinned host) on a stream that never carries a kernel, issued by the main thread
//          hipthr  the same, issued by a second host thread (the shape of the pipeline's copier thread)
//          hsa     hsa_amd_memory_async_copy (ROCr's copy entry point: the SDMA engines unless HSA_ENABLE_SDMA=0)
//          hsaeng  hsa_amd_memory_async_copy_on_engine on the engine ROCr prefers for this pair of agents
// Each mode: the copy alone and beside back-to-back HBM streaming kernels on another stream, for 1 MB and 23 MB (the
// keypoint and descriptor blocks of a batch of eight 1080p images).  Run under `rocprofv3 --kernel-trace --stats` the
// kernel list shows whether the runtime turned the copy into a blit kernel (__amd_rocclr_copyBuffer).
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define HCHECK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { printf("HSA error %d line %d\n", (int)s_, __LINE__); exit(1); } } while (0)

__global__ void stream_kernel(const float4* a, float4* b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static hsa_agent_t g_gpu, g_cpu;
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_status_t pick(hsa_agent_t a, void*) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "hip";
  const size_t DB = 1u << 30;
  float4 *da, *db;
  char *dsrc, *hdst;
  CHECK(hipMalloc(&da, DB));
  CHECK(hipMalloc(&db, DB));
  CHECK(hipMalloc(&dsrc, 32u << 20));
  CHECK(hipHostMalloc(&hdst, 32u << 20, hipHostMallocDefault));
  CHECK(hipMemset(da, 1, DB));
  CHECK(hipMemset(dsrc, 2, 32u << 20));
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  HCHECK(hsa_init());
  HCHECK(hsa_iterate_agents(pick, nullptr));
  hsa_signal_t sig;
  HCHECK(hsa_signal_create(1, 0, nullptr, &sig));
  uint32_t engine_mask = 0;
  int engine_pick = -1;  // "hsaengN": engine bit N
  if (!strncmp(mode, "hsaeng", 6)) {
    HCHECK(hsa_amd_memory_get_preferred_copy_engine(g_cpu, g_gpu, &engine_mask));
    uint32_t avail = 0;
    hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &avail);
    printf("preferred engine mask for device->host: 0x%x, free engines 0x%x\n", engine_mask, avail);
    if (mode[6]) { engine_pick = atoi(mode + 6); mode = "hsaeng"; }
  }
  auto copy = [&](size_t bytes) {
    if (!strcmp(mode, "hip")) {
      CHECK(hipMemcpyAsync(hdst, dsrc, bytes, hipMemcpyDeviceToHost, s2));
      CHECK(hipStreamSynchronize(s2));
    } else if (!strcmp(mode, "hipthr")) {
      std::thread t([&] {
        CHECK(hipSetDevice(0));
        CHECK(hipMemcpyAsync(hdst, dsrc, bytes, hipMemcpyDeviceToHost, s2));
        CHECK(hipStreamSynchronize(s2));
      });
      t.join();
    } else {
      hsa_signal_store_relaxed(sig, 1);
      if (!strcmp(mode, "hsa")) {
        HCHECK(hsa_amd_memory_async_copy(hdst, g_cpu, dsrc, g_gpu, bytes, 0, nullptr, sig));
      } else {
        const uint32_t e = engine_pick >= 0 ? (1u << engine_pick) : (engine_mask & (~engine_mask + 1));  // given, or the lowest preferred engine
        hsa_status_t st_ = hsa_amd_memory_async_copy_on_engine(hdst, g_cpu, dsrc, g_gpu, bytes, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)e, false);
        if (st_ != HSA_STATUS_SUCCESS) { printf("engine 0x%x: copy refused (status %d)\n", e, (int)st_); exit(0); }
      }
      hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
    }
  };
  auto hbm = [&]() { hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, s1, da, db, DB / 16); };
  // HBM kernel alone
  double hbm_alone = 0;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipDeviceSynchronize());
    const double t0 = now_ms();
    for (int i = 0; i < 8; i++) hbm();
    CHECK(hipStreamSynchronize(s1));
    hbm_alone = (now_ms() - t0) / 8;
  }
  printf("mode %s: hbm kernel alone %.3f ms (%.0f GB/s r+w)\n", mode, hbm_alone, 2.0 * DB / (hbm_alone * 1e-3) / 1e9);
  for (size_t bytes : {(size_t)1 << 20, (size_t)23 << 20}) {
    for (int with_hbm = 0; with_hbm < 2; with_hbm++) {
      double best_copy = 1e9, hbm_ms = 0;
      for (int rep = 0; rep < 4; rep++) {
        CHECK(hipDeviceSynchronize());
        const double t0 = now_ms();
        if (with_hbm) for (int i = 0; i < 16; i++) hbm();   // ~7 ms of streaming kernels on stream 1
        const double c0 = now_ms();
        const int n = 8;
        for (int i = 0; i < n; i++) copy(bytes);
        const double c1 = now_ms();
        CHECK(hipStreamSynchronize(s1));
        hbm_ms = (now_ms() - t0) / 16;
        if ((c1 - c0) / n < best_copy) best_copy = (c1 - c0) / n;
      }
      printf("  %5.1f MB %s: copy %.3f ms (%.1f GB/s)%s", bytes / 1048576.0, with_hbm ? "beside the hbm kernels" : "alone               ",
             best_copy, bytes / (best_copy * 1e-3) / 1e9, with_hbm ? "" : "\n");
      if (with_hbm) printf(", hbm kernel %.3f ms (%+.1f %%)\n", hbm_ms, (hbm_ms / hbm_alone - 1) * 100);
    }
  }
  return 0;
}
