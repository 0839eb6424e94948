// fetch_calib.hip -- what FETCH_SIZE (rocprofv3 --pmc FETCH_SIZE) reports for streaming reads of 4, 8 and 16 bytes per
// lane: the extrema scan reads 8 bytes per lane (float2), the guide's "x2 on gfx950" correction is calibrated on 16.
//   hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
// Each kernel reads 512 MiB once (grid-stride, 2048 x 256 threads); bytes actually read / (FETCH_SIZE x 1024) is the
// factor to apply to that access width.
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void read_kernel(const T* __restrict__ src, float* out, size_t n) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = src[i];
    s += *reinterpret_cast<const float*>(&v);
  }
  if (s == 12345.f) out[0] = s;
}

int main() {
  const size_t bytes = (size_t)512 << 20;
  void* src;
  float* out;
  if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  hipMemset(src, 1, bytes);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(read_kernel<float>, dim3(2048), dim3(256), 0, 0, (const float*)src, out, bytes / 4);
    hipLaunchKernelGGL(read_kernel<float2>, dim3(2048), dim3(256), 0, 0, (const float2*)src, out, bytes / 8);
    hipLaunchKernelGGL(read_kernel<float4>, dim3(2048), dim3(256), 0, 0, (const float4*)src, out, bytes / 16);
  }
  hipDeviceSynchronize();
  printf("read 512 MiB per launch with 4, 8 and 16 bytes per lane\n");
  return 0;
}
