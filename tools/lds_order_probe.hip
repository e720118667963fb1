// tools/lds_order_probe.hip — does an LDS atomic-add-with-return serve the lanes of ONE wavefront
// instruction in ascending lane order when several lanes name the same word?  (The architecture
// documents do not promise it.)  Every lane adds 1 to a pseudo-random word of a small table and
// compares what it got back with  base[word] + (lower lanes of this instruction on the same word).
// Other wavefronts of the workgroup hammer other LDS words meanwhile so that the timing varies.
//   hipcc --offload-arch=gfx950 -O2 -o lds_order_probe tools/lds_order_probe.hip && ./lds_order_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void probe(unsigned long long* bad, unsigned long long* total, int iters, int slots_log2) {
  __shared__ uint32_t tab[1024];
  __shared__ uint32_t noise[4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = 0;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) noise[i] = 0;
  __syncthreads();
  uint32_t rng = 0x9E3779B9u * (blockIdx.x * 1024 + threadIdx.x + 1);
  unsigned long long nbad = 0, ntot = 0;
  if (wave == 0) {
    const uint32_t mask = (1u << slots_log2) - 1u;
    for (int it = 0; it < iters; ++it) {
      rng = rng * 1664525u + 1013904223u;
      const uint32_t a = (rng >> 11) & mask;
      const uint32_t before = tab[a];                       // nobody else writes tab: stable until my wave's atomic
      __builtin_amdgcn_wave_barrier();
      const uint32_t got = __hip_atomic_fetch_add(&tab[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __builtin_amdgcn_wave_barrier();
      // lower lanes on the same word
      uint32_t lower = 0;
      for (int l = 0; l < 64; ++l) {
        const uint32_t al = (uint32_t)__builtin_amdgcn_readlane((int)a, l);
        lower += (l < lane && al == a) ? 1u : 0u;
      }
      nbad += got != before + lower ? 1 : 0;
      ntot += 1;
    }
  } else {
    for (int it = 0; it < iters * 4; ++it) {
      rng = rng * 1664525u + 1013904223u;
      atomicAdd(&noise[(rng >> 9) & 4095u], 1u);
    }
  }
  atomicAdd(bad, nbad);
  atomicAdd(total, ntot);
}

int main() {
  unsigned long long *d, h[2];
  hipMalloc(&d, 16);
  int rc = 0;
  for (int slots_log2 = 0; slots_log2 <= 10; slots_log2 += 2) {
    for (int waves = 1; waves <= 4; waves += 3) {
      hipMemset(d, 0, 16);
      hipLaunchKernelGGL(probe, dim3(1024), dim3(64 * waves), 0, 0, d, d + 1, 20000, slots_log2);
      hipDeviceSynchronize();
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("slots %4d waves %d: lane-ops %llu out of lane order %llu\n", 1 << slots_log2, waves, h[1], h[0]);
      rc |= h[0] != 0;
    }
  }
  printf(rc ? "NOT lane-ordered\n" : "lane-ordered in every case\n");
  return rc;
}
