// experiments/micro/mall_reads.hip — does a re-read that fits the 256 MB Infinity Cache (MALL) run faster than one from HBM?
// 1024 workgroups x 256 lanes stream a buffer of the given size with 16-byte loads, 20 passes per launch (the first pass of a
// launch fills the cache, the other 19 find it there if it fits).  MEASUREMENT ONLY.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void rd(const uint4* buf, int64_t n16, int passes, uint32_t* sink) {
  uint32_t acc = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int p = 0; p < passes; ++p)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * 4) {
      uint4 a = buf[i], b = i + stride < n16 ? buf[i + stride] : a, c = i + 2 * stride < n16 ? buf[i + 2 * stride] : a, d = i + 3 * stride < n16 ? buf[i + 3 * stride] : a;
      acc += a.x ^ b.y ^ c.z ^ d.w;
    }
  if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
  const size_t maxb = (size_t)4 << 30;
  uint4* buf; uint32_t* sink;
  CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, maxb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t mbs[] = {32, 64, 128, 192, 256, 384, 512, 1024, 4096};
  for (size_t mb : mbs) {
    const int64_t n16 = (int64_t)(mb << 20) / 16;
    const int passes = 20;
    hipLaunchKernelGGL(rd, dim3(1024), dim3(256), 0, 0, buf, n16, 2, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rd, dim3(1024), dim3(256), 0, 0, buf, n16, passes, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%5zu MB re-read %d times: %8.3f ms  %7.1f GB/s\n", mb, passes, ms, (double)(mb << 20) * passes / (ms * 1e-3) / 1e9);
  }
  return 0;
}
