// tests/asm/relax_instances.hip — every instance of the relaxation-form order kernel and nothing else, for
// tests/test_async_loads_asm.py: hipcc -S of this file (seconds; the whole library takes minutes), then
// tools/check_async_loads.py over the assembly.  TEST INFRASTRUCTURE: the product builds csrc/kas_hip.hip.
#define KAS_ABI_FN __host__ __device__ static inline
#include <hip/hip_runtime.h>

#include "kas_abi.h"
#include "kas_plan_math.h"
#include "kas_solver_body.h"

template <int W, bool DUAL, bool CTX, bool VERIFY, bool C16 = false, bool IDL = false, bool M32 = false, bool QUAD = false>
__global__ __launch_bounds__(64) void kas_order_relax_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::order_relax<W, DUAL, CTX, VERIFY, C16, IDL, false, M32, QUAD>(a, (int32_t)blockIdx.x, kas_lds);
}
// quad tiles (KAS_PLAN_RELAX_TILES(3)): four rows' dwords asked for per lane and step
template __global__ void kas_order_relax_kernel<3, true, false, false, false, true, true, true>(KasLaunch);
// dword mid rows (KAS_FLAG_MID32, round 6: what the headline launches): one asynchronous dword load per row
template __global__ void kas_order_relax_kernel<3, false, false, false, false, true, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, true, false, false, false, true, true>(KasLaunch);
// int32 cells, broker ids gathered from the node table (broker counts whose ids do not fit the LDS; no sampled verification)
template __global__ void kas_order_relax_kernel<2, false, false, false>(KasLaunch);
template __global__ void kas_order_relax_kernel<2, false, true, false>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, false, false, false>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, false, true, false>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, true, false, false>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, true, true, false>(KasLaunch);
// int32 cells, broker ids in the LDS (round 6: what every BASELINE config at RF 3 launches): mid rows still come in by async loads
template __global__ void kas_order_relax_kernel<2, false, false, false, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<2, false, false, true, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<2, false, true, false, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<2, false, true, true, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, false, false, false, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, false, false, true, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, false, true, false, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, false, true, true, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, true, false, false, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, true, false, true, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, true, true, false, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, true, true, true, false, true>(KasLaunch);
// the instances for 16-bit cells (kas_plan_create16), with and without the sampled verification
template __global__ void kas_order_relax_kernel<2, false, false, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<2, false, false, true, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<2, false, true, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<2, false, true, true, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, false, false, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, false, false, true, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, false, true, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, false, true, true, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, true, false, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, true, false, true, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, true, true, false, true>(KasLaunch);
template __global__ void kas_order_relax_kernel<3, true, true, true, true>(KasLaunch);
// first fit + relaxation form in one workgroup (kas_p4_order_kernel): the order wavefront's loads are the same asynchronous ones
template <int W, bool DUAL, bool C16, bool IDL, bool M32 = false, bool QUAD = false>
__global__ __launch_bounds__(128) void kas_p4_order_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::p4_order_scenario<W, DUAL, C16, IDL, M32, QUAD>(a, (int32_t)blockIdx.x, kas_lds);
}
template __global__ void kas_p4_order_kernel<3, true, false, true, true, true>(KasLaunch);
template __global__ void kas_p4_order_kernel<3, false, false, true, true>(KasLaunch);
template __global__ void kas_p4_order_kernel<3, true, false, true, true>(KasLaunch);
template __global__ void kas_p4_order_kernel<2, false, true, false>(KasLaunch);
template __global__ void kas_p4_order_kernel<2, false, false, true>(KasLaunch);
template __global__ void kas_p4_order_kernel<3, false, true, false>(KasLaunch);
template __global__ void kas_p4_order_kernel<3, false, false, true>(KasLaunch);
template __global__ void kas_p4_order_kernel<3, true, true, false>(KasLaunch);
template __global__ void kas_p4_order_kernel<3, true, false, true>(KasLaunch);
