// tools/wave_placement_probe.hip — on which SIMDs of a CU do the four wavefronts of a 256-thread workgroup land, alone and
// while long-lived one-wavefront workgroups (the order kernel's shape) sit on the SIMDs?  MEASUREMENT TOOLING.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/wave_placement_probe tools/wave_placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

// a workgroup of four wavefronts that keeps `spin` iterations busy; every wavefront records HW_REG_HW_ID and HW_REG_XCC_ID
__global__ __launch_bounds__(256) void wg4(unsigned int* rec, int spin) {
  __shared__ unsigned int pad[8192];                      // 32 KB of LDS: four such workgroups per CU, like the fill kernel
  unsigned int hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned int x = threadIdx.x;
  for (int i = 0; i < spin; ++i) { x = x * 1664525u + 1013904223u; pad[(x >> 8) & 8191u] = x; }
  if ((threadIdx.x & 63u) == 0) { rec[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; rec[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = (xcc & 15u) | (pad[x & 8191u] & 0u); }
}

// the background: one-wavefront workgroups that live long
__global__ __launch_bounds__(64) void wg1(unsigned int* sink, int spin) {
  __shared__ unsigned int pad[1280];                      // 5 KB of LDS
  unsigned int x = threadIdx.x + blockIdx.x;
  for (int i = 0; i < spin; ++i) { x = x * 1664525u + 1013904223u; pad[(x >> 8) % 1280u] = x; }
  if (x == 0xdeadbeefu) sink[0] = pad[3];
}

static void report(const char* what, const std::vector<unsigned int>& h, int n_wg) {
  int distinct4 = 0, two_on_one = 0, split_cu = 0;
  for (int b = 0; b < n_wg; ++b) {
    int simd_count[4] = {0, 0, 0, 0};
    unsigned cu0 = 0xffffffffu; bool same_cu = true;
    for (int w = 0; w < 4; ++w) {
      const unsigned hw = h[(size_t)(b * 4 + w) * 2], xcc = h[(size_t)(b * 4 + w) * 2 + 1];
      simd_count[(hw >> 4) & 3u] += 1;
      const unsigned cu = (xcc << 16) | ((hw >> 8) & 0xffu);   // cu_id, sh_id, se_id bits 8..15
      if (cu0 == 0xffffffffu) cu0 = cu; else if (cu != cu0) same_cu = false;
    }
    int mx = 0; for (int s = 0; s < 4; ++s) mx = simd_count[s] > mx ? simd_count[s] : mx;
    distinct4 += mx == 1; two_on_one += mx >= 2; split_cu += !same_cu;
  }
  printf("%-46s %6d workgroups: %6d with one wavefront on each SIMD, %6d with two or more on one SIMD, %d not on one CU\n",
         what, n_wg, distinct4, two_on_one, split_cu);
}

int main() {
  const int n_wg = 4096;
  unsigned int *d_rec, *d_sink;
  hipMalloc((void**)&d_rec, 8 * 4 * (size_t)n_wg); hipMalloc((void**)&d_sink, 64);
  std::vector<unsigned int> h(8 * (size_t)n_wg);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipLaunchKernelGGL(wg4, dim3(n_wg), dim3(256), 0, s1, d_rec, 2000);
  hipStreamSynchronize(s1);
  hipMemcpy(h.data(), d_rec, 4 * h.size(), hipMemcpyDeviceToHost);
  report("four-wavefront workgroups alone", h, n_wg);
  // background first (1 per SIMD: 1024 one-wavefront workgroups that outlive the probe), then the probe
  for (int per_simd = 1; per_simd <= 3; ++per_simd) {
    hipLaunchKernelGGL(wg1, dim3(1024 * per_simd), dim3(64), 0, s2, d_sink, 4000000);
    hipLaunchKernelGGL(wg4, dim3(n_wg), dim3(256), 0, s1, d_rec, 2000);
    hipStreamSynchronize(s1);
    hipMemcpy(h.data(), d_rec, 4 * h.size(), hipMemcpyDeviceToHost);
    char what[96]; snprintf(what, sizeof(what), "with %d long one-wavefront workgroups per SIMD", per_simd);
    report(what, h, n_wg);
    hipStreamSynchronize(s2);
  }
  return 0;
}
