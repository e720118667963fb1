// tools/issue_probe.hip — what does a gfx950 SIMD issue per cycle?  MEASUREMENT TOOLING, not a product path.
//
// The solver kernels of this repository are chains of integer VALU / SALU / LDS instructions, and DESIGN.md section 4.5
// prices them against an issue ceiling.  This probe measures that ceiling instead of quoting it: per instruction kind,
// W = 1, 2, 4, 8 wavefronts per SIMD (blocks of 256 threads = one wave per SIMD, LDS-padded so that exactly W blocks
// are resident per CU, grid = CUs x W: every wave is resident from the start), each wave running ITER x 128
// instructions of that kind over 8 independent registers.  Printed per row:
//   cyc/inst/wave   s_memtime ticks of a wave / instructions it issued (what ONE wave sees)
//   inst/cyc/SIMD   W x instructions / ticks (what the SIMD sustains) -> the pipe's rate when it stops growing with W
//   clock           s_memtime ticks / wall time of the launch (HIP events)
//   hipcc --offload-arch=gfx950 -O2 -o tools/issue_probe tools/issue_probe.hip && tools/issue_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define R8(a) a(0) a(1) a(2) a(3) a(4) a(5) a(6) a(7)
#define X16(s) s s s s s s s s s s s s s s s s

#define OPS "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)

enum Kind {
  K_ADD, K_AND, K_LSHL, K_CNDMASK, K_BFE, K_MIN3, K_LSHL_OR, K_ADD3, K_MUL_LO, K_MAD_U64, K_LSHL_B64, K_CMP, K_MBCNT,
  K_READLANE, K_BCNT, K_ADD_DEP, K_SALU, K_MIX_VS, K_MIX_VSS, K_DS_READ, K_DS_READ_RANDOM, K_DS_ADD, K_DS_ADD_RANDOM, K_DS_ADD_RTN64, K_BPERMUTE, K_MIX_VL,
  K_CND_SGPR, K_CND_CONST, K_MOV, K_XOR, K_SUB, K_MAX, K_LSHR, K_OR3, K_AND_OR, K_BFI, K_PERM, K_MED3, K_BITOP3, K_LSHL_ADD, K_LSHL_ADD_U64,
  K_ADD_CO, K_MAD_U24, K_MUL_U24, K_CMP_VCC, K_SAVEEXEC, K_WRITELANE, K_MOV_HALF_EXEC, K_CND_DPP, K_CNDMASK_AFTER_ADD,
  K_COUNT
};
static const char* kind_name[K_COUNT] = {
  "v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_cndmask_b32", "v_bfe_u32", "v_min3_u32", "v_lshl_or_b32", "v_add3_u32",
  "v_mul_lo_u32", "v_mad_u64_u32", "v_lshlrev_b64", "v_cmp_lt_u32 (sgpr pair)", "v_mbcnt_lo_u32_b32", "v_readlane_b32",
  "v_bcnt_u32_b32", "v_add_u32 dependent chain", "s_add_u32", "v_add_u32 + s_add_u32 alternating", "v_add_u32 + 2 x s_add_u32",
  "ds_read_b32 lane-linear", "ds_read_b32 random word of 1024", "ds_add_u32 lane-linear", "ds_add_u32 random word of 1024",
  "ds_add_rtn_u64 random word of 1024", "ds_bpermute_b32", "7 x v_add_u32 + ds_read_b32 random",
  "v_cndmask_b32_e64 (sgpr pair mask)", "v_cndmask_b32 vcc, constants 0 / 1", "v_mov_b32", "v_xor_b32", "v_sub_u32", "v_max_u32", "v_lshrrev_b32", "v_or3_b32", "v_and_or_b32",
  "v_bfi_b32", "v_perm_b32", "v_med3_u32", "v_bitop3_b32", "v_lshl_add_u32", "v_lshl_add_u64", "v_add_co_u32 + v_addc_co_u32", "v_mad_u32_u24", "v_mul_u32_u24",
  "v_cmp_eq_u32 vcc", "s_and_saveexec_b64 + s_mov exec", "v_writelane_b32", "v_mov_b32 under half exec", "v_mov_b32_dpp row_shr:1", "v_cndmask_b32 vcc, 1 in 8 (7 v_add_u32 between)",
};

template <int KIND>
__global__ __launch_bounds__(512) void probe(uint32_t* sink, unsigned long long* ticks, int iters) {
  extern __shared__ uint32_t lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += blockDim.x) lds[i] = i;
  __syncthreads();
  uint32_t r0 = tid, r1 = tid + 1, r2 = tid + 2, r3 = tid + 3, r4 = tid + 4, r5 = tid + 5, r6 = tid + 6, r7 = tid + 7;
  uint32_t k = 3 + (tid & 1);
  uint32_t rnd = (uint32_t)((tid * 2654435761u) >> 20) & 1023u;
  uint32_t lin = (uint32_t)tid * 4u, rnd4 = rnd * 4u, rnd8 = (rnd & 511u) * 8u, bp = (uint32_t)((tid * 7) & 63) * 4u;
  uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4;
  unsigned long long q0 = tid, q1 = tid + 9;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == K_ADD) {
#define L(i) "v_add_u32 %" #i ", %" #i ", %8\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k));
#undef L
    } else if constexpr (KIND == K_AND) {
#define L(i) "v_and_b32 %" #i ", %" #i ", %8\n"
      asm volatile(X16(R8(L)) : OPS : "v"(~k));
#undef L
    } else if constexpr (KIND == K_LSHL) {
#define L(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
      asm volatile(X16(R8(L)) : OPS);
#undef L
    } else if constexpr (KIND == K_CNDMASK) {
#define L(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
      asm volatile("v_cmp_gt_u32 vcc, 3, %8\n" X16(R8(L)) : OPS : "v"(k) : "vcc");
#undef L
    } else if constexpr (KIND == K_BFE) {
#define L(i) "v_bfe_u32 %" #i ", %" #i ", 1, 20\n"
      asm volatile(X16(R8(L)) : OPS);
#undef L
    } else if constexpr (KIND == K_MIN3) {
#define L(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k), "v"(rnd));
#undef L
    } else if constexpr (KIND == K_LSHL_OR) {
#define L(i) "v_lshl_or_b32 %" #i ", %" #i ", 2, %8\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k));
#undef L
    } else if constexpr (KIND == K_ADD3) {
#define L(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k), "v"(rnd));
#undef L
    } else if constexpr (KIND == K_MUL_LO) {
#define L(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k));
#undef L
    } else if constexpr (KIND == K_MAD_U64) {
      // 4 independent 64-bit accumulators
      asm volatile(X16("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
                       "v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                       "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
                       "v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n")
                   : "+v"(q0), "+v"(q1), "+v"(*(unsigned long long*)&r0), "+v"(*(unsigned long long*)&r2) : "v"(k), "v"(rnd) : "vcc");
    } else if constexpr (KIND == K_LSHL_B64) {
      asm volatile(X16("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n"
                       "v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n")
                   : "+v"(q0), "+v"(q1));
    } else if constexpr (KIND == K_CMP) {
      asm volatile(X16("v_cmp_lt_u32 s[20:21], %0, %8\n v_cmp_lt_u32 s[22:23], %1, %8\n v_cmp_lt_u32 s[24:25], %2, %8\n v_cmp_lt_u32 s[26:27], %3, %8\n"
                       "v_cmp_lt_u32 s[20:21], %4, %8\n v_cmp_lt_u32 s[22:23], %5, %8\n v_cmp_lt_u32 s[24:25], %6, %8\n v_cmp_lt_u32 s[26:27], %7, %8\n")
                   : OPS : "v"(k) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if constexpr (KIND == K_MBCNT) {
#define L(i) "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k));
#undef L
    } else if constexpr (KIND == K_READLANE) {
      asm volatile(X16("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n"
                       "v_readlane_b32 s24, %4, 3\n v_readlane_b32 s25, %5, 5\n v_readlane_b32 s26, %6, 7\n v_readlane_b32 s27, %7, 9\n")
                   : OPS : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if constexpr (KIND == K_BCNT) {
#define L(i) "v_bcnt_u32_b32 %" #i ", %8, %" #i "\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k));
#undef L
    } else if constexpr (KIND == K_ADD_DEP) {
      asm volatile(X16("v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n"
                       "v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n") : OPS : "v"(k));
    } else if constexpr (KIND == K_SALU) {
      asm volatile(X16("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n"
                       "s_add_u32 %0, %0, 5\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 5\n")
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
    } else if constexpr (KIND == K_MIX_VS) {
      asm volatile(X16("v_add_u32 %0, %0, %8\n s_add_u32 s20, s20, 3\n v_add_u32 %1, %1, %8\n s_add_u32 s21, s21, 3\n"
                       "v_add_u32 %2, %2, %8\n s_add_u32 s22, s22, 3\n v_add_u32 %3, %3, %8\n s_add_u32 s23, s23, 3\n")
                   : OPS : "v"(k) : "scc", "s20", "s21", "s22", "s23");
    } else if constexpr (KIND == K_MIX_VSS) {
      asm volatile(X16("v_add_u32 %0, %0, %8\n s_add_u32 s20, s20, 3\n s_add_u32 s21, s21, 3\n v_add_u32 %1, %1, %8\n s_add_u32 s22, s22, 3\n s_add_u32 s23, s23, 3\n"
                       "v_add_u32 %2, %2, %8\n s_add_u32 s20, s20, 3\n")
                   : OPS : "v"(k) : "scc", "s20", "s21", "s22", "s23");
    } else if constexpr (KIND == K_DS_READ) {
#define L(i) "ds_read_b32 %" #i ", %8\n"
      asm volatile(X16(R8(L)) "s_waitcnt lgkmcnt(0)\n" : OPS : "v"(lin) : "memory");
#undef L
    } else if constexpr (KIND == K_DS_READ_RANDOM) {
#define L(i) "ds_read_b32 %" #i ", %8\n"
      asm volatile(X16(R8(L)) "s_waitcnt lgkmcnt(0)\n" : OPS : "v"(rnd4) : "memory");
#undef L
    } else if constexpr (KIND == K_DS_ADD) {
#define L(i) "ds_add_u32 %8, %" #i "\n"
      asm volatile(X16(R8(L)) "s_waitcnt lgkmcnt(0)\n" : OPS : "v"(lin) : "memory");
#undef L
    } else if constexpr (KIND == K_DS_ADD_RANDOM) {
#define L(i) "ds_add_u32 %8, %" #i "\n"
      asm volatile(X16(R8(L)) "s_waitcnt lgkmcnt(0)\n" : OPS : "v"(rnd4) : "memory");
#undef L
    } else if constexpr (KIND == K_DS_ADD_RTN64) {
      asm volatile(X16("ds_add_rtn_u64 %0, %2, %0\n ds_add_rtn_u64 %1, %2, %1\n ds_add_rtn_u64 %0, %2, %0\n ds_add_rtn_u64 %1, %2, %1\n"
                       "ds_add_rtn_u64 %0, %2, %0\n ds_add_rtn_u64 %1, %2, %1\n ds_add_rtn_u64 %0, %2, %0\n ds_add_rtn_u64 %1, %2, %1\n")
                   "s_waitcnt lgkmcnt(0)\n" : "+v"(q0), "+v"(q1) : "v"(rnd8) : "memory");
    } else if constexpr (KIND == K_BPERMUTE) {
#define L(i) "ds_bpermute_b32 %" #i ", %8, %" #i "\n"
      asm volatile(X16(R8(L)) "s_waitcnt lgkmcnt(0)\n" : OPS : "v"(bp) : "memory");
#undef L
    } else if constexpr (KIND == K_MIX_VL) {
      asm volatile(X16("ds_read_b32 %7, %9\n v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                       "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n")
                   "s_waitcnt lgkmcnt(0)\n" : OPS : "v"(k), "v"(rnd4) : "memory");

    } else if constexpr (KIND == K_CND_SGPR) {
#define L(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
      asm volatile("v_cmp_gt_u32 s[20:21], 3, %8\n" X16(R8(L)) : OPS : "v"(k) : "s20", "s21");
#undef L
    } else if constexpr (KIND == K_CND_CONST) {
#define L(i) "v_cndmask_b32 %" #i ", 0, %8, vcc\n"
      asm volatile("v_cmp_gt_u32 vcc, 3, %8\n" X16(R8(L)) : OPS : "v"(k) : "vcc");
#undef L
    } else if constexpr (KIND == K_MOV) {
#define L(i) "v_mov_b32 %" #i ", %8\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k));
#undef L
    } else if constexpr (KIND == K_XOR) {
#define L(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k));
#undef L
    } else if constexpr (KIND == K_SUB) {
#define L(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k));
#undef L
    } else if constexpr (KIND == K_MAX) {
#define L(i) "v_max_u32 %" #i ", %" #i ", %8\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k));
#undef L
    } else if constexpr (KIND == K_LSHR) {
#define L(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
      asm volatile(X16(R8(L)) : OPS);
#undef L
    } else if constexpr (KIND == K_OR3) {
#define L(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k), "v"(rnd));
#undef L
    } else if constexpr (KIND == K_AND_OR) {
#define L(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k), "v"(rnd));
#undef L
    } else if constexpr (KIND == K_BFI) {
#define L(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k), "v"(rnd));
#undef L
    } else if constexpr (KIND == K_PERM) {
#define L(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k), "v"(rnd));
#undef L
    } else if constexpr (KIND == K_MED3) {
#define L(i) "v_med3_u32 %" #i ", %" #i ", %8, %9\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k), "v"(rnd));
#undef L
    } else if constexpr (KIND == K_BITOP3) {
#define L(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k), "v"(rnd));
#undef L
    } else if constexpr (KIND == K_LSHL_ADD) {
#define L(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k));
#undef L
    } else if constexpr (KIND == K_LSHL_ADD_U64) {
      asm volatile(X16("v_lshl_add_u64 %0, %0, 1, %2\n v_lshl_add_u64 %1, %1, 1, %2\n v_lshl_add_u64 %0, %0, 1, %2\n v_lshl_add_u64 %1, %1, 1, %2\n"
                       "v_lshl_add_u64 %0, %0, 1, %2\n v_lshl_add_u64 %1, %1, 1, %2\n v_lshl_add_u64 %0, %0, 1, %2\n v_lshl_add_u64 %1, %1, 1, %2\n")
                   : "+v"(q0), "+v"(q1) : "v"(*(unsigned long long*)&r0));
    } else if constexpr (KIND == K_ADD_CO) {
      asm volatile(X16("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n"
                       "v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n")
                   : OPS : "v"(k) : "vcc");
    } else if constexpr (KIND == K_MAD_U24) {
#define L(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k), "v"(rnd));
#undef L
    } else if constexpr (KIND == K_MUL_U24) {
#define L(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k));
#undef L
    } else if constexpr (KIND == K_CMP_VCC) {
#define L(i) "v_cmp_eq_u32 vcc, %" #i ", %8\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k) : "vcc");
#undef L
    } else if constexpr (KIND == K_SAVEEXEC) {
      asm volatile("v_cmp_gt_u32 vcc, 3, %8\n s_mov_b64 s[22:23], exec\n"
                   X16("s_and_saveexec_b64 s[20:21], vcc\n s_mov_b64 exec, s[20:21]\n s_and_saveexec_b64 s[20:21], vcc\n s_mov_b64 exec, s[20:21]\n"
                       "s_and_saveexec_b64 s[20:21], vcc\n s_mov_b64 exec, s[20:21]\n s_and_saveexec_b64 s[20:21], vcc\n s_mov_b64 exec, s[20:21]\n")
                   "s_mov_b64 exec, s[22:23]\n" : OPS : "v"(k) : "vcc", "s20", "s21", "s22", "s23", "scc");
    } else if constexpr (KIND == K_WRITELANE) {
#define L(i) "v_writelane_b32 %" #i ", s20, 3\n"
      asm volatile("s_mov_b32 s20, 5\n" X16(R8(L)) : OPS : : "s20");
#undef L
    } else if constexpr (KIND == K_MOV_HALF_EXEC) {
#define L(i) "v_mov_b32 %" #i ", %8\n"
      asm volatile("v_cmp_gt_u32 vcc, 4, %8\n s_and_saveexec_b64 s[20:21], vcc\n" X16(R8(L)) "s_mov_b64 exec, s[20:21]\n" : OPS : "v"(k) : "vcc", "s20", "s21", "scc");
#undef L
    } else if constexpr (KIND == K_CND_DPP) {
#define L(i) "v_mov_b32_dpp %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      asm volatile(X16(R8(L)) : OPS : "v"(k));
#undef L
    } else if constexpr (KIND == K_CNDMASK_AFTER_ADD) {
      asm volatile("v_cmp_gt_u32 vcc, 3, %8\n"
                   X16("v_cndmask_b32 %0, %0, %8, vcc\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                       "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n") : OPS : "v"(k) : "vcc");
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((tid & 63) == 0) ticks[blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + tid] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + s0 + s1 + s2 + s3 + (uint32_t)q0 + (uint32_t)q1;
}

// does ds_add_rtn_u64 serve the lanes of one instruction in lane order (same question as lds_order_probe.hip asks for u32)?
__global__ void order64(unsigned long long* bad, unsigned long long* total, int iters, int slots_log2) {
  __shared__ unsigned long long tab[1024];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = 0;
  __syncthreads();
  uint32_t rng = 0x9E3779B9u * (blockIdx.x * 1024 + threadIdx.x + 1);
  unsigned long long nbad = 0, ntot = 0;
  if (threadIdx.x < 64) {
    const uint32_t mask = (1u << slots_log2) - 1u;
    for (int it = 0; it < iters; ++it) {
      rng = rng * 1664525u + 1013904223u;
      const uint32_t a = (rng >> 11) & mask;
      const unsigned long long before = tab[a];
      __builtin_amdgcn_wave_barrier();
      const unsigned long long got = __hip_atomic_fetch_add(&tab[a], (1ull << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __builtin_amdgcn_wave_barrier();
      uint32_t lower = 0;
      for (int l = 0; l < 64; ++l) {
        const uint32_t al = (uint32_t)__builtin_amdgcn_readlane((int)a, l);
        lower += (l < lane && al == a) ? 1u : 0u;
      }
      nbad += got != before + (((unsigned long long)lower << 32) | lower) ? 1 : 0;
      ntot += 1;
    }
  } else {
    // the other waves keep the LDS busy with conflicting traffic on their own words
    __shared__ uint32_t noise[2048];
    for (int it = 0; it < iters * 2; ++it) {
      rng = rng * 1664525u + 1013904223u;
      atomicAdd(&noise[(rng >> 9) & 2047u], 1u);
    }
  }
  atomicAdd(bad, nbad);
  atomicAdd(total, ntot);
}

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int KIND>
static void run(int cus, uint32_t* sink, unsigned long long* ticks, int iters) {
  for (int W = 1; W <= 8; W *= 2) {
    // LDS padding: exactly W waves per SIMD resident (blocks of 256 threads = one wave per SIMD; W = 8: 4 blocks of 512)
    const int per_cu = W <= 4 ? W : 4, threads = W <= 4 ? 256 : 512;
    const int lds = (160 * 1024) / per_cu - 1024;
    HIP_OK(hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = cus * per_cu;
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(threads), lds, 0, sink, ticks, 8);      // warm
    HIP_OK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(threads), lds, 0, sink, ticks, iters);
    HIP_OK(hipEventRecord(e1, 0));
    HIP_OK(hipDeviceSynchronize());
    float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * (threads / 64));
    HIP_OK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0, mx = 0;
    for (auto v : h) { sum += (double)v; if ((double)v > mx) mx = (double)v; }
    const double avg = sum / h.size();
    const double insts = (double)iters * 128.0;
    printf("%-48s W=%d  cyc/inst/wave %6.2f  (slowest wave %6.2f -> cyc/inst/SIMD %5.2f)  launch %.3f ms  ticks/us %.0f\n",
           kind_name[KIND], W, avg / insts, mx / insts, mx / insts / W, ms, mx / (ms * 1000.0));
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  hipDeviceProp_t prop; HIP_OK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  uint32_t* sink; unsigned long long* ticks;
  HIP_OK(hipMalloc(&sink, (size_t)cus * 8 * 512 * 4));
  HIP_OK(hipMalloc(&ticks, (size_t)cus * 8 * 4 * 8));
  run<K_ADD>(cus, sink, ticks, iters);
  run<K_AND>(cus, sink, ticks, iters);
  run<K_LSHL>(cus, sink, ticks, iters);
  run<K_CNDMASK>(cus, sink, ticks, iters);
  run<K_BFE>(cus, sink, ticks, iters);
  run<K_MIN3>(cus, sink, ticks, iters);
  run<K_LSHL_OR>(cus, sink, ticks, iters);
  run<K_ADD3>(cus, sink, ticks, iters);
  run<K_MUL_LO>(cus, sink, ticks, iters);
  run<K_MAD_U64>(cus, sink, ticks, iters);
  run<K_LSHL_B64>(cus, sink, ticks, iters);
  run<K_CMP>(cus, sink, ticks, iters);
  run<K_MBCNT>(cus, sink, ticks, iters);
  run<K_READLANE>(cus, sink, ticks, iters);
  run<K_BCNT>(cus, sink, ticks, iters);
  run<K_ADD_DEP>(cus, sink, ticks, iters);
  run<K_SALU>(cus, sink, ticks, iters);
  run<K_MIX_VS>(cus, sink, ticks, iters);
  run<K_MIX_VSS>(cus, sink, ticks, iters);
  run<K_DS_READ>(cus, sink, ticks, iters);
  run<K_DS_READ_RANDOM>(cus, sink, ticks, iters);
  run<K_DS_ADD>(cus, sink, ticks, iters);
  run<K_DS_ADD_RANDOM>(cus, sink, ticks, iters);
  run<K_DS_ADD_RTN64>(cus, sink, ticks, iters);
  run<K_BPERMUTE>(cus, sink, ticks, iters);
  run<K_MIX_VL>(cus, sink, ticks, iters);
  run<K_CND_SGPR>(cus, sink, ticks, iters);
  run<K_CND_CONST>(cus, sink, ticks, iters);
  run<K_CNDMASK_AFTER_ADD>(cus, sink, ticks, iters);
  run<K_MOV>(cus, sink, ticks, iters);
  run<K_MOV_HALF_EXEC>(cus, sink, ticks, iters);
  run<K_CND_DPP>(cus, sink, ticks, iters);
  run<K_XOR>(cus, sink, ticks, iters);
  run<K_SUB>(cus, sink, ticks, iters);
  run<K_MAX>(cus, sink, ticks, iters);
  run<K_LSHR>(cus, sink, ticks, iters);
  run<K_OR3>(cus, sink, ticks, iters);
  run<K_AND_OR>(cus, sink, ticks, iters);
  run<K_BFI>(cus, sink, ticks, iters);
  run<K_PERM>(cus, sink, ticks, iters);
  run<K_MED3>(cus, sink, ticks, iters);
  run<K_BITOP3>(cus, sink, ticks, iters);
  run<K_LSHL_ADD>(cus, sink, ticks, iters);
  run<K_LSHL_ADD_U64>(cus, sink, ticks, iters);
  run<K_ADD_CO>(cus, sink, ticks, iters);
  run<K_MAD_U24>(cus, sink, ticks, iters);
  run<K_MUL_U24>(cus, sink, ticks, iters);
  run<K_CMP_VCC>(cus, sink, ticks, iters);
  run<K_SAVEEXEC>(cus, sink, ticks, iters);
  run<K_WRITELANE>(cus, sink, ticks, iters);
  // lane order of 64-bit LDS atomics with return
  unsigned long long *d, h[2];
  HIP_OK(hipMalloc(&d, 16));
  int rc = 0;
  for (int slots_log2 = 0; slots_log2 <= 10; slots_log2 += 2) {
    for (int waves = 1; waves <= 4; waves += 3) {
      HIP_OK(hipMemset(d, 0, 16));
      hipLaunchKernelGGL(order64, dim3(1024), dim3(64 * waves), 0, 0, d, d + 1, 5000, slots_log2);
      HIP_OK(hipDeviceSynchronize());
      HIP_OK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
      printf("ds_add_rtn_u64 order: slots %4d waves %d: lane-ops %llu out of lane order %llu\n", 1 << slots_log2, waves, h[1], h[0]);
      rc |= h[0] != 0;
    }
  }
  return rc;
}
