// Shared device helpers of the 64-row half-block kernels (fused_wide2.hip: forward; bwd_wide2.hip: backward, first half): MFMA fragment
// types, the weight-stream stage (buffer loads with scalar offsets), per-register constant vectors, the developer timeline stamp and
// the late kernel-argument pointer loads.  Everything sits in an anonymous namespace: include once per translation unit.
#pragma once
#include "fused_rows.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 join(s16x4 lo, s16x4 hi) { return bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}; }
__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = v;
  return z;
}
__device__ __forceinline__ float bf_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return __uint_as_float(v & 0xFFFF0000u); }
// accumulator register r of lane half h holds row (r & 3) + 8 (r >> 2) + 4 h of the 32x32 tile
__device__ __forceinline__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ u32x4 pack8(const f32x16& a, int k) {
  return u32x4{pack2(a[8 * k], a[8 * k + 1]), pack2(a[8 * k + 2], a[8 * k + 3]), pack2(a[8 * k + 4], a[8 * k + 5]), pack2(a[8 * k + 6], a[8 * k + 7])};
}
__device__ __forceinline__ void store16_wt(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// Developer timeline (testing hook "stamps"): lane 0 of every wave records the 100 MHz wall clock at phase boundaries:
// stamps[(block * 8 + wave) * 16 + k] (the 8-wave kernels' layout: waves 4..7 of a block stay empty here).
__device__ __forceinline__ void stamp(unsigned long long* stamps, int k) {
  if (stamps && (threadIdx.x & 63) == 0) {
    unsigned long long* p = stamps + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16;
    p[k] = __builtin_amdgcn_s_memrealtime();
    if (k == 0) p[11] = __builtin_amdgcn_s_memtime();
    if (k == 12) p[15] = __builtin_amdgcn_s_memtime();
    // (the block's XCD, in the first slot of the unused wave row 4: HW_REG_XCC_ID = hardware register 20, bits 3..0)
    if (k == 0 && threadIdx.x == 0) stamps[((size_t)blockIdx.x * 8 + 4) * 16] = 1 + (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15);
  }
}

constexpr int NW = 4;        // waves per block
constexpr int NTH = 64 * NW;
constexpr int RT = 2;        // 32-row sub-tiles per block
constexpr int ROWS = 32 * RT;
constexpr int PX = 272;      // row pitch (bytes) of a [rows][128] bf16 tile
constexpr int PR = 528;      // ... of a [rows][256] bf16 tile (a ds_read_b128 lane group's 16 rows land on 16 distinct bank quads)
constexpr int PS = 144;      // row pitch (bytes) of a strip [rows][64]: 16 rows of a ds_read_b128 lane group land on 16 distinct 16-byte slots
// KG->RG partial of one (segment, head), this file's layout inside the workspace's partial buffer (sized for the other kernels'
// FUSED_PART_FLOATS = 544 floats per slot): max[16] fp32 (log2 units) | sum[16] fp32 | Z[16 queries][32 features] BF16 -- the partials
// are 150 KB per sample in fp32 and the KG rows' launch reads all of them at once (160 MB at B = 1024: its combine ran at the HBM
// roof); Z is a sum of <= 64 products whose factors are bf16 already, and the attention output it ends in is rounded to bf16 too
constexpr int PART_FLOATS = 16 + 16 + 16 * 32 / 2;
static_assert(PART_FLOATS <= FUSED_PART_FLOATS, "fused_rows.h");
// training calls keep Z in fp32 (the other kernels' slot layout: max[16] | sum[16] | Z[16][32] fp32): the backward's gradients pass
// through ReLU decisions of the 13 KG rows, and the bf16-operand oracle that bounds them models fp32 partial sums
constexpr int PART_FLOATS_F32 = FUSED_PART_FLOATS;
constexpr float LOG2E = 1.4426950408889634f;
// one-instruction transcendentals (v_exp_f32, v_rcp_f32, v_rsq_f32: 1 ulp): exp2f / division / 1 / sqrtf compile to range-fixing and
// Newton sequences of 5-15 instructions each, which the bf16 operands downstream cannot see
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }

// ---- one linear layer: acc[s][t] = init[t] + sum over KS k steps of (weight fragment (ks, t)) x (activation fragment (s, ks)).
// `wp` = this wave's first fragment + lane (16-byte units); fragment (ks, t) is wp[64 (ks KST + t)] (KST = tiles per k step and
// wave in the shadow's 4-wave layout).  DEPTH weight fragments are kept in flight; the scheduling barriers pin {issue the load
// DEPTH fragments ahead, (first tile of a k step: start the NEXT k step's activation reads), RS MFMAs} -- left alone, hipcc sinks
// loads and LDS reads behind the MFMAs and every k step waits out a round trip (fused_wide.hip, StageW).
template <int RS, int NT, int KS, int KST, int DEPTH>
struct Stage {
  static constexpr int TOTAL = NT * KS;
  static constexpr int D = DEPTH < TOTAL ? DEPTH : TOTAL;
  u32x4 buf[D];
  // weight fragments come through buffer loads: a wave-uniform descriptor and byte offset in SGPRs (advanced by scalar adds), ONE
  // per-lane offset register for the whole kernel -- as 64-bit per-lane pointers every other fragment cost two vector adds
  __amdgpu_buffer_rsrc_t rs;
  int voff, soff;
  __device__ __forceinline__ u32x4 load(int i) const {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + 1024 * ((i / NT) * KST + (i % NT)), 0);
  }
  // base: the shadow (wave-uniform pointer), frag0: this wave's first fragment (wave-uniform index, 1 KB units)
  __device__ __forceinline__ void prefetch(const us16* base, int frag0, int lane) {
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<us16*>(base), 0, 0x7FFFFFFF, 0x00020000);
    voff = 16 * lane; soff = 1024 * frag0;
#pragma unroll
    for (int i = 0; i < D; ++i) buf[i] = load(i);
  }
  // W_IS_A: the weights are the A operand (accumulator: lane = tile row, registers = features) -- else the B operand (lane = feature,
  // registers = tile rows).  init[t]: the first k step's C operand (a bias vector, or zeros).
  template <bool W_IS_A, class F>
  __device__ __forceinline__ void run_f(F&& frag, const f32x16 (&init)[NT], f32x16 (&acc)[RS][NT]) {
    bf16x8 x[RS], xn[RS];
#pragma unroll
    for (int s = 0; s < RS; ++s) xn[s] = frag(s, 0);
#pragma unroll
    for (int i = 0; i < TOTAL; ++i) {
      const int ks = i / NT, t = i % NT;
      const bf16x8 wf = as_frag(buf[i % D]);
      if (i + D < TOTAL) buf[i % D] = load(i + D);
      if (t == 0) {
#pragma unroll
        for (int s = 0; s < RS; ++s) x[s] = xn[s];
        if (ks + 1 < KS) {
#pragma unroll
          for (int s = 0; s < RS; ++s) xn[s] = frag(s, ks + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < RS; ++s) {
        const f32x16 c = ks == 0 ? init[t] : acc[s][t];
        if constexpr (W_IS_A) acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, x[s], c, 0, 0, 0);
        else                  acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[s], wf, c, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // run_f with a C operand per (sub-tile, feature tile): initf(s, t) (the backward's first product starts from a per-sample vector)
  template <bool W_IS_A, class F, class I>
  __device__ __forceinline__ void run_fi(F&& frag, I&& initf, f32x16 (&acc)[RS][NT]) {
    bf16x8 x[RS], xn[RS];
#pragma unroll
    for (int s = 0; s < RS; ++s) xn[s] = frag(s, 0);
#pragma unroll
    for (int i = 0; i < TOTAL; ++i) {
      const int ks = i / NT, t = i % NT;
      const bf16x8 wf = as_frag(buf[i % D]);
      if (i + D < TOTAL) buf[i % D] = load(i + D);
      if (t == 0) {
#pragma unroll
        for (int s = 0; s < RS; ++s) x[s] = xn[s];
        if (ks + 1 < KS) {
#pragma unroll
          for (int s = 0; s < RS; ++s) xn[s] = frag(s, ks + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < RS; ++s) {
        const f32x16 c = ks == 0 ? initf(s, t) : acc[s][t];
        if constexpr (W_IS_A) acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, x[s], c, 0, 0, 0);
        else                  acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[s], wf, c, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // activation tiles in LDS: `act` = address of this lane's first fragment of sub-tile 0 (row lane & 31, byte 16 (lane >> 5)),
  // sub-tile s is `sub` bytes further, k step ks 32 bytes further
  template <bool W_IS_A>
  __device__ __forceinline__ void run(const char* act, int sub, const f32x16 (&init)[NT], f32x16 (&acc)[RS][NT]) {
    run_f<W_IS_A>([&](int s, int ks) { return *reinterpret_cast<const bf16x8*>(act + s * sub + 32 * ks); }, init, acc);
  }
};

// a per-register constant vector for the W_IS_A orientation: register i = c[acc_row(i, h)], c = 32 floats in LDS (four 16-byte reads)
__device__ __forceinline__ f32x16 feature_vec(const float* c, int h) {
  f32x16 v;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(c + 8 * g + 4 * h);
    v[4 * g] = q[0]; v[4 * g + 1] = q[1]; v[4 * g + 2] = q[2]; v[4 * g + 3] = q[3];
  }
  return v;
}

struct Sub { int b; int row0; int nr; float inv_n; };          // one 32-row sub-tile (wave-uniform)

// A pointer of the kernel's argument block re-read from the kernarg segment AT ITS POINT OF USE (the block is the kernel's only
// argument: byte offset = offsetof in it).  The training variants write nine saved tensors, each through its own 64-bit pointer, once;
// held in SGPRs from the kernel's entry those pointers overflow the scalar file and come back as v_readlane / v_writelane traffic all
// over the kernel.  The opaque offset keeps hipcc from hoisting the load back to the entry.
template <class T>
__device__ __forceinline__ T* karg(int byte_off) {
  asm volatile("" : "+s"(byte_off));
  typedef const char __attribute__((address_space(4))) kchar;
  typedef T* const __attribute__((address_space(4))) kptr;
  kchar* ka = (kchar*)__builtin_amdgcn_kernarg_segment_ptr();
  return *(kptr*)(ka + byte_off);
}
#define KOFF(Args, path) ((int)__builtin_offsetof(Args, path))
#define LATEP(cond, T, off, expr) ((cond) ? karg<T>(off) : (expr))      // the training variants (cond) take the pointer late

}  // namespace
