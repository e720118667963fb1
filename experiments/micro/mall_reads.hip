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
// a plain copy (16-byte loads and stores, half of the bytes each way): what HBM delivers to mixed traffic
__global__ __launch_bounds__(256) void cp(const uint4* in, uint4* out, int64_t n16) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * 4) {
    uint4 a = in[i], b = i + stride < n16 ? in[i + stride] : a, c = i + 2 * stride < n16 ? in[i + 2 * stride] : a, d = i + 3 * stride < n16 ? in[i + 3 * stride] : a;
    out[i] = a;
    if (i + stride < n16) out[i + stride] = b;
    if (i + 2 * stride < n16) out[i + 2 * stride] = c;
    if (i + 3 * stride < n16) out[i + 3 * stride] = d;
  }
}
// two thirds reads, one third writes (the job's mix: 63 % reads): out[i] = in[i] ^ in2[i]
__global__ __launch_bounds__(256) void cp2(const uint4* in, const uint4* in2, uint4* out, int64_t n16) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * 2) {
    uint4 a = in[i], a2 = in2[i];
    const bool two = i + stride < n16;
    uint4 b = two ? in[i + stride] : a, b2 = two ? in2[i + stride] : a2;
    out[i] = make_uint4(a.x ^ a2.x, a.y ^ a2.y, a.z ^ a2.z, a.w ^ a2.w);
    if (two) out[i + stride] = make_uint4(b.x ^ b2.x, b.y ^ b2.y, b.z ^ b2.z, b.w ^ b2.w);
  }
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
  {
    const int64_t n16 = (int64_t)((size_t)1 << 30) / 16;      // 1 GB in, 1 GB out (two halves of the buffer)
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(cp, dim3(2048), dim3(256), 0, 0, buf, buf + n16, n16);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("copy 1 GB -> 1 GB: %8.3f ms per launch  %7.1f GB/s of reads + writes\n", ms / 5, 2.0 * (1 << 30) / (ms / 5 * 1e-3) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(cp2, dim3(2048), dim3(256), 0, 0, buf, buf + n16, buf + 2 * n16, n16);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("2 x 1 GB in -> 1 GB out: %8.3f ms per launch  %7.1f GB/s of reads + writes\n", ms / 5, 3.0 * (1 << 30) / (ms / 5 * 1e-3) / 1e9);
    }
  }
  return 0;
}
