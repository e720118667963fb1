#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <chrono>
__global__ void spin(float* out, int iters, unsigned* ids) {
  float x = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
  if (x == 12345.f) out[0] = x;
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    ids[blockIdx.x] = (hw & 0xffffu) | (xcc << 16);
  }
}
static unsigned* d_ids; static float* d;
static double run(hipStream_t s, int blocks, int iters = 200000) {
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, d, iters, d_ids);
  hipStreamSynchronize(s);
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
static void who(int blocks, const char* name) {
  std::vector<unsigned> h(blocks);
  hipMemcpy(h.data(), d_ids, 4 * blocks, hipMemcpyDeviceToHost);
  std::set<unsigned> cus; int per_xcc[8] = {0};
  for (unsigned v : h) {
    const unsigned cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7, xcc = (v >> 16) & 0xf;
    if (cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu).second) per_xcc[xcc & 7]++;
  }
  printf("   %s: %zu distinct CUs; per XCC:", name, cus.size());
  for (int i = 0; i < 8; ++i) printf(" %d", per_xcc[i]);
  printf("\n");
}
static std::vector<uint32_t> range(int lo, int hi) { std::vector<uint32_t> m(8, 0); for (int i = lo; i < hi; ++i) m[i / 32] |= 1u << (i % 32); return m; }
int main() {
  hipMalloc(&d, 4); hipMalloc(&d_ids, 4 * 8192);
  hipStream_t s0; hipStreamCreate(&s0);
  run(s0, 2048);
  printf("full: %.2f ms\n", run(s0, 2048)); who(2048, "full");
  struct { const char* name; std::vector<uint32_t> m; } v[] = {
    {"bits 0..31", range(0, 32)}, {"bits 0..95", range(0, 96)}, {"bits 96..255", range(96, 256)},
    {"even bits", std::vector<uint32_t>(8, 0x55555555u)}, {"bits 0..7", range(0, 8)}, {"low 16 of each 32", std::vector<uint32_t>(8, 0x0000ffffu)}};
  for (auto& e : v) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, e.m.data()) != hipSuccess) { printf("%s: create failed\n", e.name); continue; }
    run(s, 2048);
    printf("%s: %.2f ms\n", e.name, run(s, 2048)); who(2048, e.name);
    hipStreamDestroy(s);
  }
  auto a = range(0, 96), b = range(96, 256);
  hipStream_t sa, sb;
  hipExtStreamCreateWithCUMask(&sa, 8, a.data());
  hipExtStreamCreateWithCUMask(&sb, 8, b.data());
  run(sa, 768); run(sb, 1280);
  printf("A alone (0..95) 768 blocks: %.2f ms; ", run(sa, 768)); printf("B alone (96..255) 1280 blocks: %.2f ms\n", run(sb, 1280));
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(spin, dim3(768), dim3(256), 0, sa, d, 200000, d_ids);
  hipLaunchKernelGGL(spin, dim3(1280), dim3(256), 0, sb, d, 200000, d_ids + 4096);
  hipDeviceSynchronize();
  printf("A and B together: %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  return 0;
}
