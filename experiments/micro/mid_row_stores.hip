// experiments/micro/mid_row_stores.hip — how much does the SHAPE of a 6-byte-per-row store / load cost on gfx950?
// The fill's second scan stores one mid row per lane as a dword + a halfword at a 6-byte stride (packed rows); the order kernel
// loads them the same way.  Variants, one wavefront per 64 rows, 4 wavefronts per workgroup, every variant moves 6 bytes per row:
//   0  packed:  dword at 6 p, halfword at 6 p + 4                       (the product)
//   1  planar:  per tile of 64 rows 64 dwords, then 64 halfwords        (both instructions contiguous and aligned)
//   2  staged:  rows through 384 B of LDS per wavefront, then 24 lanes x 16 bytes
// and the same three as loads (3..5).  MEASUREMENT ONLY: hipcc --offload-arch=gfx950 -O3 -o mid_row_stores mid_row_stores.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>

__device__ inline void store_u32_a2(uint16_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ inline uint32_t load_u32_a2(const uint16_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

template <int MODE>
__global__ __launch_bounds__(256) void k(uint16_t* buf, int64_t rows_per_wg, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) uint16_t stg[4][192];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wg_row0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t tiles = rows_per_wg / 64 / 4;               // per wavefront
  uint16_t* base = buf + (wg_row0 + (int64_t)wave * tiles * 64) * 3;
  uint32_t acc = 0;
  for (int64_t t = 0; t < tiles; ++t) {
    uint16_t* tile = base + t * 192;
    const uint32_t a = (uint32_t)(t * 64 + lane) * 2654435761u;
    const uint32_t h01 = a, h2 = a >> 7;
    if (MODE == 0) {
      store_u32_a2(tile + lane * 3, h01);
      tile[lane * 3 + 2] = (uint16_t)h2;
    } else if (MODE == 1) {
      reinterpret_cast<uint32_t*>(tile)[lane] = h01;
      tile[128 + lane] = (uint16_t)h2;
    } else if (MODE == 2) {
      stg[wave][lane * 3] = (uint16_t)h01; stg[wave][lane * 3 + 1] = (uint16_t)(h01 >> 16); stg[wave][lane * 3 + 2] = (uint16_t)h2;
      __builtin_amdgcn_wave_barrier();
      if (lane < 24) reinterpret_cast<uint4*>(tile)[lane] = reinterpret_cast<const uint4*>(stg[wave])[lane];
      __builtin_amdgcn_wave_barrier();
    } else if (MODE == 3) {
      acc += load_u32_a2(tile + lane * 3) + tile[lane * 3 + 2];
    } else if (MODE == 4) {
      acc += reinterpret_cast<const uint32_t*>(tile)[lane] + tile[128 + lane];
    } else if (MODE == 5) {
      if (lane < 24) reinterpret_cast<uint4*>(stg[wave])[lane] = reinterpret_cast<const uint4*>(tile)[lane];
      __builtin_amdgcn_wave_barrier();
      acc += stg[wave][lane * 3] + stg[wave][lane * 3 + 1] + stg[wave][lane * 3 + 2];
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (MODE >= 3 && acc == 0x12345678u) sink[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
  const int wgs = 1000;
  const int64_t rows_per_wg = 100000 / 256 * 256;           // ~100k rows per workgroup, like a scenario of the headline
  const size_t bytes = (size_t)wgs * rows_per_wg * 6;
  uint16_t* buf; uint32_t* sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[6] = {"store packed (dword + halfword at a 6-byte stride)", "store planar (64 dwords, 64 halfwords)", "store staged through LDS (24 lanes x 16 B)",
                          "load packed", "load planar", "load staged through LDS"};
  for (int rep = 0; rep < 2; ++rep)
    for (int m = 0; m < 6; ++m) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 10; ++i) {
        switch (m) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, buf, rows_per_wg, sink); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, buf, rows_per_wg, sink); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, buf, rows_per_wg, sink); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), 0, 0, buf, rows_per_wg, sink); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(256), 0, 0, buf, rows_per_wg, sink); break;
          default: hipLaunchKernelGGL(k<5>, dim3(wgs), dim3(256), 0, 0, buf, rows_per_wg, sink); break;
        }
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-60s %8.3f ms per launch  %7.1f GB/s\n", names[m], ms / 10, bytes / (ms / 10 * 1e-3) / 1e9);
    }
  return 0;
}
