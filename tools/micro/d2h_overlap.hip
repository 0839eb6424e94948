// Does device->host traffic disturb a kernel that streams HBM?  Times, alone and together on two streams:
//   (a) hipMemcpyAsync device->pinned host of 16 MB,  (b) a shader copy HBM -> pinned host of 16 MB (64 workgroups),
//   (c) an HBM->HBM streaming kernel over 1 GB.
//   hipcc --offload-arch=gfx950 -O2 -o d2h_overlap d2h_overlap.hip && ./d2h_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void stream_kernel(const float4* a, float4* b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

// Paced shader copy to the host: a wavefront stores K KB, then waits until those stores are acknowledged before the
// next K KB, so at most waves * K KB are in flight towards PCIe.
template <int K>
__global__ void paced_copy_kernel(const float4* a, float4* b, size_t n) {
  const size_t waves = (size_t)gridDim.x * (blockDim.x / 64);
  const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int lane = threadIdx.x & 63;
  for (size_t base = wave * 64 * K; base < n; base += waves * 64 * K) {
    float4 v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = base + k * 64 + lane < n ? a[base + k * 64 + lane] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < K; k++) if (base + k * 64 + lane < n) b[base + k * 64 + lane] = v[k];
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): loads and stores of this wavefront have completed
  }
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  const size_t HB = 16u << 20, DB = 1u << 30;
  float4 *da, *db, *dsrc, *hdst;
  hipMalloc(&da, DB); hipMalloc(&db, DB); hipMalloc(&dsrc, HB);
  hipHostMalloc(&hdst, HB, hipHostMallocDefault);
  hipMemset(da, 1, DB); hipMemset(dsrc, 2, HB);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  printf("HSA_ENABLE_SDMA=%s\n", getenv("HSA_ENABLE_SDMA") ? getenv("HSA_ENABLE_SDMA") : "(unset)");
  auto hbm = [&]() { hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, s1, da, db, DB / 16); };
  auto dma = [&]() { hipMemcpyAsync(hdst, dsrc, HB, hipMemcpyDeviceToHost, s2); };
  auto shader = [&]() { hipLaunchKernelGGL(stream_kernel, dim3(64), dim3(256), 0, s2, dsrc, hdst, HB / 16); };
  for (int mode = 0; mode < 5; mode++) {
    const char* names[] = {"hbm kernel alone", "dma copy alone", "shader copy alone", "hbm kernel + dma copy", "hbm kernel + shader copy"};
    for (int rep = 0; rep < 3; rep++) {
      hipDeviceSynchronize();
      const double t0 = now_ms();
      const int n = 8;
      for (int i = 0; i < n; i++) {
        if (mode == 0 || mode >= 3) hbm();
        if (mode == 1 || mode == 3) dma();
        if (mode == 2 || mode == 4) shader();
      }
      hipStreamSynchronize(s1);
      const double t1 = now_ms();
      hipStreamSynchronize(s2);
      const double t2 = now_ms();
      if (rep == 2)
        printf("%-26s stream1 done after %.3f ms (%.0f GB/s HBM r+w), all done after %.3f ms (%.1f GB/s to host)\n", names[mode],
               (t1 - t0) / n, (mode == 0 || mode >= 3) ? 2.0 * DB / ((t1 - t0) / n * 1e-3) / 1e9 : 0.0, (t2 - t0) / n,
               (mode != 0) ? HB / ((t2 - t0) / n * 1e-3) / 1e9 : 0.0);
    }
  }
  // paced shader copies beside the HBM kernel
  {
    struct Cfg { int blocks, threads, k; } cfgs[] = {{8, 256, 4}, {16, 256, 4}, {32, 256, 4}, {64, 256, 4}, {16, 256, 8}, {32, 256, 8}, {64, 256, 1}, {256, 256, 1}};
    for (auto& cf : cfgs) {
      for (int with_hbm = 0; with_hbm < 2; with_hbm++) {
        double best1 = 1e9, best2 = 1e9;
        for (int rep = 0; rep < 3; rep++) {
          hipDeviceSynchronize();
          const double t0 = now_ms();
          const int n = 8;
          for (int i = 0; i < n; i++) {
            if (with_hbm) hbm();
            if (cf.k == 1) hipLaunchKernelGGL(paced_copy_kernel<1>, dim3(cf.blocks), dim3(cf.threads), 0, s2, dsrc, hdst, HB / 16);
            else if (cf.k == 4) hipLaunchKernelGGL(paced_copy_kernel<4>, dim3(cf.blocks), dim3(cf.threads), 0, s2, dsrc, hdst, HB / 16);
            else hipLaunchKernelGGL(paced_copy_kernel<8>, dim3(cf.blocks), dim3(cf.threads), 0, s2, dsrc, hdst, HB / 16);
          }
          hipStreamSynchronize(s1);
          const double t1 = now_ms();
          hipStreamSynchronize(s2);
          const double t2 = now_ms();
          best1 = (t1 - t0) / n < best1 ? (t1 - t0) / n : best1;
          best2 = (t2 - t0) / n < best2 ? (t2 - t0) / n : best2;
        }
        printf("paced copy %3d blocks x %d waves x %d KB in flight (%4d KB)%s: hbm kernel %.3f ms, copy done %.3f ms (%.1f GB/s)\n", cf.blocks,
               cf.threads / 64, cf.k, cf.blocks * cf.threads / 64 * cf.k, with_hbm ? " + hbm kernel" : "             ", with_hbm ? best1 : 0.0, best2,
               HB / (best2 * 1e-3) / 1e9);
      }
    }
  }
  // Does hipMemcpyAsync(device->host) behind a kernel return to the host at once?
  for (int variant = 0; variant < 2; variant++) {
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    hipDeviceSynchronize();
    const double t0 = now_ms();
    for (int i = 0; i < 4; i++) hbm();  // ~2 ms of kernels on stream 1
    const double t1 = now_ms();
    if (variant == 0) {
      hipMemcpyAsync(hdst, dsrc, HB, hipMemcpyDeviceToHost, s1);
    } else {
      hipEventRecord(ev, s1);
      hipStreamWaitEvent(s2, ev, 0);
      hipMemcpyAsync(hdst, dsrc, HB, hipMemcpyDeviceToHost, s2);
    }
    const double t2 = now_ms();
    hipDeviceSynchronize();
    const double t3 = now_ms();
    printf("%s: kernels enqueued in %.3f ms, copy call returned after %.3f ms, everything done after %.3f ms\n",
           variant == 0 ? "copy on the kernels' stream" : "copy on a second stream behind an event", t1 - t0, t2 - t1, t3 - t0);
  }
  return 0;
}
