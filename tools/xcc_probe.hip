// tools/xcc_probe.hip — where do the workgroups of a kernel launched on a CU-masked stream run?
// MEASUREMENT TOOLING (not a product path).  For a few mask patterns: per XCD (HW_REG_XCC_ID) the number of workgroups
// and of distinct CUs (HW_REG_HW_ID: se / sh / cu) that took them.  Build: hipcc --offload-arch=gfx950 -O2 -o tools/xcc_probe tools/xcc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ __launch_bounds__(64) void probe(unsigned int* rec, int spin) {
  unsigned int xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // keep the CU busy for a while so that the launch spreads over everything it may use
  unsigned int x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;
  if (threadIdx.x == 0) { rec[2 * blockIdx.x] = xcc; rec[2 * blockIdx.x + 1] = hw ^ (x & 0u); }
}

static void run(const char* name, const std::vector<uint32_t>& mask, int n_wg) {
  hipStream_t s;
  hipError_t e = mask.empty() ? hipStreamCreateWithFlags(&s, hipStreamNonBlocking)
                              : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%-34s stream: %s\n", name, hipGetErrorString(e)); return; }
  unsigned int* d; hipMalloc((void**)&d, 8 * (size_t)n_wg); hipMemset(d, 0xff, 8 * (size_t)n_wg);
  hipLaunchKernelGGL(probe, dim3(n_wg), dim3(64), 0, s, d, 20000);
  hipStreamSynchronize(s);
  std::vector<unsigned int> h(2 * (size_t)n_wg);
  hipMemcpy(h.data(), d, 8 * (size_t)n_wg, hipMemcpyDeviceToHost);
  int wg[16] = {0}; bool seen[16][1024]; memset(seen, 0, sizeof(seen));
  for (int i = 0; i < n_wg; ++i) {
    const unsigned xcc = h[2 * i] & 15u, hw = h[2 * i + 1];
    const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
    wg[xcc] += 1; seen[xcc][(se << 5) | (sh << 4) | cu] = true;
  }
  printf("%-34s", name);
  for (int x = 0; x < 8; ++x) { int c = 0; for (int k = 0; k < 1024; ++k) c += seen[x][k]; printf(" x%d:%5d wg/%2d cu", x, wg[x], c); }
  printf("\n");
  hipFree(d); hipStreamDestroy(s);
}

int main() {
  const int cus = 256, words = 8, n_wg = 16384;
  auto mk = [&](auto pred) { std::vector<uint32_t> m(words, 0u); for (int i = 0; i < cus; ++i) if (pred(i)) m[i >> 5] |= 1u << (i & 31); return m; };
  run("no mask", {}, n_wg);
  run("bits i%8 < 3", mk([](int i) { return i % 8 < 3; }), n_wg);
  run("bits i%8 >= 3", mk([](int i) { return i % 8 >= 3; }), n_wg);
  run("bits i%8 == 0", mk([](int i) { return i % 8 == 0; }), n_wg);
  run("bits 160..255", mk([](int i) { return i >= 160; }), n_wg);
  run("bits 0..159", mk([](int i) { return i < 160; }), n_wg);
  run("bits i%32 < 12", mk([](int i) { return i % 32 < 12; }), n_wg);
  run("bits i%32 >= 12", mk([](int i) { return i % 32 >= 12; }), n_wg);
  run("bits 0..31", mk([](int i) { return i < 32; }), n_wg);
  run("bits (i/8)%4 == 0", mk([](int i) { return (i / 8) % 4 == 0; }), n_wg);
  return 0;
}
