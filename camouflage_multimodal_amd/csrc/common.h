// Shared device/host helpers for the fusion kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CAMO_WAVE 64

// dropout sites: one independent mask stream per place the reference has an
// nn.Dropout / MHA dropout (fusion_model.py:36,44,56,63,71,211,218,225,232; :157,160)
enum : uint32_t {
  SITE_ATTN_RG2KG = 1, SITE_ATTN_KG2RG = 2, SITE_FFN_RG = 3, SITE_FFN_KG = 4, SITE_FUSE = 5,
  SITE_HEAD0 = 6, /* +0 mask, +1 instance, +2 edge, +3 score */
  SITE_LATE0 = 10 /* +0, +1 */
};

struct DropCfg {
  uint32_t seed_lo, seed_hi;
  float p;       // drop probability; 0 => disabled
  float scale;   // 1/(1-p)
  uint32_t thr;  // keep iff hash >= thr: ceil(p * 2^24) << 8, i.e. exactly "top 24 bits / 2^24 >= p" without the shift, convert and multiply
};

__host__ __device__ inline DropCfg make_drop(int training, float p, uint64_t seed) {
  DropCfg d;
  d.seed_lo = (uint32_t)(seed & 0xFFFFFFFFull);
  d.seed_hi = (uint32_t)(seed >> 32);
  d.p = (training && p > 0.f) ? p : 0.f;
  d.scale = d.p > 0.f ? 1.0f / (1.0f - d.p) : 1.0f;
  // u = (x >> 8) / 2^24 >= p  <=>  (x >> 8) >= p * 2^24 (exact in double)  <=>  x >= ceil(p * 2^24) << 8  (the low 8 bits of x cannot
  // lift x >> 8 over an integer bound)
  const double t = (double)d.p * 16777216.0;
  unsigned long long t24 = (unsigned long long)t;
  if ((double)t24 < t) ++t24;
  d.thr = t24 >= 16777216ull ? 0xFFFFFFFFu : (uint32_t)(t24 << 8);
  return d;
}

__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}

// keep-multiplier (0 or 1/(1-p)) of element `idx` at `site`.  Same arithmetic as
// oracle/fusion_oracle.py::dropout_keep, so masks agree element for element.  One murmur3 finaliser over the
// element counter offset by a per-(site, seed) constant and whitened with the seed's high word: two 32-bit
// multiplies per element (v_mul_lo_u32 is quarter rate on gfx950; the two-round version of round 1 cost ~150
// cycles per element and showed up as microseconds in every epilogue that draws a mask).
__device__ __forceinline__ float drop_mult(const DropCfg& d, uint32_t site, uint32_t idx) {
  const uint32_t x = fmix32((idx + (site * 0x85EBCA77u + d.seed_lo)) ^ d.seed_hi);
  return x >= d.thr ? d.scale : 0.0f;
}

// lane `i` of the VGPR `v` <- the wave-uniform 32-bit value `s`: one v_writelane_b32.  This hipcc declares no writelane builtin, so
// the LLVM intrinsic is bound by its name -- NOT inline assembly: gfx950 wants two wait states between a VALU instruction writing an
// SGPR (the ballot's v_cmp) and a VALU instruction reading it, which the compiler's hazard pass provides for its own instructions only.
// Used to hand the halves of 16 ballots to 16 lanes: a `lane == i` select per ballot keeps 16 compare masks and 16 ballots (64 SGPRs)
// alive instead.
extern "C" __device__ int camo_writelane_i32(int, int, int) __asm("llvm.amdgcn.writelane.i32");
#define SET_LANE(v, s, i) (v) = (uint32_t)camo_writelane_i32((int)(s), (i), (int)(v))

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over the 64 lanes with DPP adds only (no LDS crossbar round trips): a shift-scan inside each row of 16, then the two
// row broadcasts.  The total is valid in LANE 63 only.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = dpp_add<0x111, 0xf>(v);      // row_shr:1
  v = dpp_add<0x112, 0xf>(v);      // row_shr:2
  v = dpp_add<0x114, 0xf>(v);      // row_shr:4
  v = dpp_add<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of each row holds the row's sum
  v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// fp32 -> bf16 bit pattern, round-to-nearest-even (what every MFMA operand of the bf16 schedule sees)
__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;     // round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(unsigned short, b);
}
// two floats -> one dword of two bf16 (lo in bits 0..15): ONE v_cvt_pk_bf16_f32.  Written as a vector conversion: two scalar
// conversions + shift + or let the SLP vectoriser pair elements of DIFFERENT dwords and then shuffle halves back with
// v_and / v_lshl / v_or_sdwa / v_mov (13 instructions per 4 values instead of 4 in every bf16 epilogue).
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
