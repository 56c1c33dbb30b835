// Fused row-tile kernels of the bf16 schedule (gfx950); contract in fused_rows.h, layouts in tests/test_fragment_maps.py.
//
// Why this shape.  At the reference sizes a training step is ~8 k RG rows x a chain of five 256-wide linear layers: one
// 32-row tile per CU.  Run as one GEMM launch per layer, every activation makes an HBM round trip between launches and
// each launch pays its own fill/drain (round 1: 8 launches x 15-24 us at 7 % of the MFMA peak).  Here a block keeps its
// 32 rows on chip through the whole chain and only the weights move: each of the 4 waves owns a quarter of a layer's
// output features, streams exactly its quarter of the weight matrix from L2 straight into registers (the shadow copy is
// stored in MFMA-fragment order, so a wave-instruction is one contiguous 1-KB read and no weight touches LDS), and meets
// the other waves only through the 32 x 256 bf16 activation tile in LDS that the next layer reads as its other operand.
//
// Orientation.  With the weights as the MFMA "A" operand a stage computes out^T: lane = row, registers = features.  Row
// statistics (LayerNorm, softmax over the 13 keys) are then in-lane sums plus one exchange between the lane halves, the
// epilogue writes 4 consecutive features per store, and an attention head's 32 features are exactly one accumulator tile
// that feeds the next MFMA as an operand without leaving the registers (cdna_hip_programming.md, "an accumulator tile as
// the next MFMA's operand").  Stages whose epilogue reduces over ROWS (the pooled FFN activation) swap the operands:
// lane = feature, registers = rows.
#include "fused_rows.h"
#include "shadow_inl.h"
#include "gemm.h"      // launch timing hooks (gemm_prof_open / close)

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
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ float bf_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return __uint_as_float(v & 0xFFFF0000u); }
// accumulator register r of lane half h holds row (r & 3) + 8 (r >> 2) + 4 h of the 32x32 tile
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- one linear layer of a tile: acc[t] (+)= sum over KS k steps.  `wp` = the wave's fragment stream + lane (16-byte
// units: fragment i of the stream is wp[64 i]); `act` = LDS address of this lane's first activation fragment (row
// lane & 31, byte offset 16 (lane >> 5)); k step ks is 32 bytes further.  DEPTH weight fragments are kept in flight.
// `rot` (block-uniform) rotates the order of the k steps (tiles of a launch all walk the SAME weight stream).
// prefetch() issues the first DEPTH weight loads and may be called well before run() -- ahead of the epilogue or the
// attention that produces the stage's input -- so that the stream is already flowing when the MFMAs start.
template <int NT, int KS, bool W_IS_A, int DEPTH, bool USE_ROT = true>
struct Stage {
  static constexpr int TOTAL = NT * KS;
  static constexpr int D = DEPTH < TOTAL ? DEPTH : TOTAL;
  static_assert(!USE_ROT || (KS & (KS - 1)) == 0, "rotated k order: power-of-two k steps");
  u32x4 buf[D];
  const u32x4* wp; int rot;
  __device__ __forceinline__ int keff(int ks) const { return USE_ROT ? ((ks + rot) & (KS - 1)) : ks; }
  __device__ __forceinline__ const u32x4* wptr(int i) const { return wp + 64 * (keff(i / NT) * NT + (i % NT)); }
  __device__ __forceinline__ void prefetch(const u32x4* __restrict__ wp_, int rot_) {
    wp = wp_; rot = rot_;
#pragma unroll
    for (int i = 0; i < D; ++i) buf[i] = *wptr(i);
  }
  __device__ __forceinline__ void run(const char* act, f32x16 (&acc)[NT]) {
    run_f([&](int ks) { return *reinterpret_cast<const bf16x8*>(act + 32 * ks); }, acc);
  }
  // frag(ks): this lane's activation fragment of k step ks (the default reads it from the LDS tile)
  template <class F>
  __device__ __forceinline__ void run_f(F&& frag, f32x16 (&acc)[NT]) {
    bf16x8 x = frag(keff(0)), xn = x;
    // One step = {issue the weight load DEPTH fragments ahead, (first tile of a k step: start the NEXT k step's activation
    // read), MFMA}.  The scheduling barrier pins that order: left alone, hipcc sinks every load next to its use and the
    // stream runs at one L2 round trip per fragment (measured: s_waitcnt vmcnt(1) in front of almost every MFMA).
#pragma unroll
    for (int i = 0; i < TOTAL; ++i) {
      const int ks = i / NT, t = i % NT;
      const bf16x8 wf = as_frag(buf[i % D]);
      if (i + D < TOTAL) buf[i % D] = *wptr(i + D);
      if (t == 0) {
        x = xn;
        if (ks + 1 < KS) xn = frag(keff(ks + 1));
      }
      if constexpr (W_IS_A) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, x, acc[t], 0, 0, 0);
      else                  acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, wf, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// Developer timeline (testing hook "stamps"): when a buffer is given, wave 0 of every block records the 100 MHz wall clock
// at its phase boundaries: stamps[block * 8 + k].  Product calls pass null and execute none of it.
__device__ __forceinline__ void stamp(unsigned long long* stamps, int k) {
  if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memrealtime();
}

// rows [0, nrows) of a bf16 LDS tile -> global, 16 bytes per thread, whole rows contiguous (LOG2C: log2 of 16-byte chunks per row)
// 16-byte write-through store (sc0 sc1): the bytes go to memory as they are issued instead of staying dirty in this XCD's L2
// until the end-of-kernel write-back, which then has that much less to do before the next launch may start
// (tile outputs of one training step: ~60 MB; measured -2.5 us per step at B = 16)
__device__ __forceinline__ void store16_wt(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");   // (s_nop: the data registers may be rewritten right behind an asm store)
}
template <int LOG2C>
__device__ __forceinline__ void copy_out(const char* lds, int pitch, int col_byte0, us16* dst, int ld, size_t row0, int nrows) {
  for (int c = threadIdx.x; c < (32 << LOG2C); c += 256) {
    const int r = c >> LOG2C, k = c & ((1 << LOG2C) - 1);
    if (r < nrows)
      store16_wt(reinterpret_cast<char*>(dst + (row0 + r) * (size_t)ld) + 16 * k, *reinterpret_cast<const u32x4*>(lds + r * pitch + col_byte0 + 16 * k));
  }
}

// ------------------------------------------------------------------------------------------------ weight shadows
__global__ __launch_bounds__(256) void shadow_kernel(const ShadowBatch sb, int total_chunks) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  for (int zi = 0; zi < sb.nzero; ++zi) {                  // the step's atomics block and operand pad rows ride along
    u32x4* z = static_cast<u32x4*>(sb.zero_ptr[zi]);
    const size_t n16 = sb.zero_bytes[zi] >> 4;
    for (size_t i = gid; i < n16; i += (size_t)gridDim.x * 256) z[i] = u32x4{0u, 0u, 0u, 0u};
  }
  if (gid >= total_chunks) return;
  int ji = 0;
#pragma unroll
  for (int i = 1; i < SHADOW_MAXJ; ++i)
    if (i < sb.n && gid >= sb.j[i].chunk_begin) ji = i;
  shadow_chunk(sb.j[ji], gid - sb.j[ji].chunk_begin);
}

// ------------------------------------------------------------------------------------------------ forward, front half
constexpr int PX = 272;     // row pitch (bytes) of the [32][128] bf16 input tile: 256 + 16
constexpr int PR = 528;     // ... of a [32][256] bf16 tile: a ds_read_b128 lane group's 16 rows land on 16 distinct bank quads
constexpr int PQ = 1552;    // ... of the [32][768] bf16 [q | k' | v'] tile
constexpr int F_BUFR = 32 * PX, F_BUFQ = F_BUFR + 32 * PR, F_LDS = F_BUFQ + 32 * PQ;      // 8704, 25600, 75264

template <int DEPTH, bool ROT>
__global__ __launch_bounds__(256, 2) void front_kernel(const FrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FrontStream& S = a.s[(int)blockIdx.x >= a.s[1].tile_begin ? 1 : 0];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = ((int)blockIdx.x - S.tile_begin) * 32;
  const int nrows = min(32, S.M - row0);
  char* bufX = smem; char* bufR = smem + F_BUFR; char* bufQ = smem + F_BUFQ;
  const int rot = ROT ? (int)(blockIdx.x >> 3) : 0;         // (blocks b, b + 8, ... share an XCD's L2)
  stamp(a.stamps, 0);
  Stage<2, 8, true, DEPTH> st0;
  st0.prefetch(reinterpret_cast<const u32x4*>(S.W0) + (size_t)w * (8 * 2 * 64) + lane, rot);
  {   // input tile: fp32 -> bf16, rows past the end cleared
    const int r = tid >> 3, c = tid & 7;
    const float4* src = reinterpret_cast<const float4*>(S.X + (size_t)(row0 + min(r, nrows - 1)) * 128 + 16 * c);
    float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
    if (r >= nrows) { v0 = v1 = v2 = v3 = make_float4(0.f, 0.f, 0.f, 0.f); }
    *reinterpret_cast<u32x4*>(bufX + r * PX + 32 * c) = u32x4{pack2(v0.x, v0.y), pack2(v0.z, v0.w), pack2(v1.x, v1.y), pack2(v1.z, v1.w)};
    *reinterpret_cast<u32x4*>(bufX + r * PX + 32 * c + 16) = u32x4{pack2(v2.x, v2.y), pack2(v2.z, v2.w), pack2(v3.x, v3.y), pack2(v3.z, v3.w)};
  }
  __syncthreads();
  // projection 128 -> 256  (every global store of the tile waits for the end of the kernel: vmcnt counts stores too and
  // retires in order, so a store issued in front of a weight stream makes the stream's first waits sit out its write latency)
  Stage<6, 16, true, DEPTH> st1;
  {
    f32x16 acc[2] = {zero16(), zero16()};
    st0.run(bufX + l31 * PX + 16 * h, acc);
    st1.prefetch(reinterpret_cast<const u32x4*>(S.W1) + (size_t)w * (16 * 6 * 64) + lane, rot);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 64 * w + 32 * t + 8 * g + 4 * h;
        const float4 bv = *reinterpret_cast<const float4*>(S.b0 + c0);
        *reinterpret_cast<u32x2*>(bufR + l31 * PR + 2 * c0) =
            u32x2{pack2(acc[t][4 * g] + bv.x, acc[t][4 * g + 1] + bv.y), pack2(acc[t][4 * g + 2] + bv.z, acc[t][4 * g + 3] + bv.w)};
      }
  }
  __syncthreads();
  stamp(a.stamps, 1);
  // in-projections 256 -> [256 q | 512 k', v'] (24 feature tiles, 6 per wave)
  {
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) acc[t] = zero16();
    st1.run(bufR + l31 * PR + 16 * h, acc);
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const int tg = 6 * w + t;                                  // wave-uniform
      const float* bias = tg < 8 ? S.bq + 32 * tg : S.bkv + 32 * (tg - 8);
      const float sc = tg < 8 ? a.qscale : 1.0f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 8 * g + 4 * h;
        const float4 bv = *reinterpret_cast<const float4*>(bias + c);
        *reinterpret_cast<u32x2*>(bufQ + l31 * PQ + 2 * (32 * tg + c)) =
            u32x2{pack2((acc[t][4 * g] + bv.x) * sc, (acc[t][4 * g + 1] + bv.y) * sc),
                  pack2((acc[t][4 * g + 2] + bv.z) * sc, (acc[t][4 * g + 3] + bv.w) * sc)};
      }
    }
  }
  __syncthreads();
  stamp(a.stamps, 2);
  copy_out<5>(bufQ, PQ, 0, S.Q16, 256, row0, nrows);
  copy_out<6>(bufQ, PQ, 512, S.KV16, 512, row0, nrows);
  copy_out<5>(bufR, PR, 0, S.R16, 256, row0, nrows);
  if (a.save) copy_out<4>(bufX, PX, 0, S.X16, 128, row0, nrows);
  for (int zi = 0; zi < a.nzero; ++zi) {                    // (behind the tile's own stores: nothing of this block waits for them)
    u32x4* z = static_cast<u32x4*>(a.zero_ptr[zi]);
    const unsigned n16 = a.zero_bytes[zi] >> 4;
    for (unsigned i = blockIdx.x * 256 + tid; i < n16; i += gridDim.x * 256) z[i] = u32x4{0u, 0u, 0u, 0u};
  }
  stamp(a.stamps, 3);
}

// ------------------------------------------------------------------------------------------------ forward, back half
// LDS map (bytes).  Region A is the attention scratch and, once the attention and the out-projection are done with it, the
// fp32 tile of the LayerNorm output whose columns are summed for the mean pool.
constexpr int PK = 528;      // RG tile: the sample's 16 key rows [16][256] (ds_read_b128 fragments: rows 4 banks apart)
constexpr int PV = 576;      // RG tile: the sample's 16 value rows [16][256] (transposing reads: rows 16 banks apart)
constexpr int PVC = 192;     // KG block: a wave's 32-row x 64-feature value chunk (transposing reads)
constexpr int PT = 260;      // fp32 tile pitch (floats)
constexpr int B_VS = 16 * PK;                       // 8448
// Two layouts.  OCC = 2 (launches of at most two blocks per CU: small batches, one tile per CU is the critical path): region A =
// [0, 33280) is the attention scratch (8448 + 9216 bytes for an RG tile, 4 * 32 * 192 for a KG block) and, once the attention and the
// out-projection are done with it, the fp32 tile of the LayerNorm output whose columns are summed for the mean pool (only the
// tile's real rows are added: 13 for a KG chain); 68 KB.  OCC = 3 (larger launches): the pool is a halving butterfly in registers, the
// LayerNorm output tile (bufY) and the normalised LayerNorm input of a saving call (bufXH) take region A's place; 51.7 KB, three
// blocks per CU, 168 VGPRs.
template <int OCC> struct BackL;
template <> struct BackL<2> { static constexpr int XH = 0, O = 32 * PT * 4, Y = O + 32 * PR, RED = Y + 32 * PR, LDS = RED + 1024; };       // 0, 33280, 50176, 67072, 68096
template <> struct BackL<3> { static constexpr int Y = 0, XH = 32 * PR, O = 2 * 32 * PR, RED = O + 32 * PR, LDS = RED + 1024; };             // 0, 16896, 33792, 50688, 51712
constexpr int B_LDS = BackL<2>::LDS;
static_assert(BackL<3>::O >= 4 * 32 * 192 && BackL<3>::O >= 8448 + 9216 && BackL<2>::O >= 4 * 32 * 192, "attention scratch inside region A");

// Softmax over the <= 16 keys of one RG row: S holds the (pre-scaled) scores of keys acc_row(i, h), i < 8, in this lane
// and the other 8 keys in lane ^ 32.  p = probabilities (0 for keys >= Nk).  Forward and backward run this same code.
__device__ __forceinline__ void rg_softmax(const f32x16& S, int h, int Nk, float (&p)[8]) {
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 8; ++i) { p[i] = acc_row(i, h) < Nk ? S[i] : -INFINITY; m = fmaxf(m, p[i]); }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { p[i] = __expf(p[i] - m); sum += p[i]; }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] *= inv;
}

// RG tile: rows [row0, row0 + nrows) of sample b against its Nk keys; wave w owns heads 2w, 2w+1.  Leaves the attention
// output (bf16) in bufO.
__device__ __forceinline__ void attn_rg_tile(const BackArgs& a, char* smem, char* bufO, int b, size_t row0, int nrows, int w, int lane) {
  const int tid = threadIdx.x, l31 = lane & 31, h = lane >> 5, Nk = a.Nk;
  char* Ks = smem; char* Vs = smem + B_VS;
  // the sample's key | value rows: [Nk][512] bf16 -> two images, rows Nk..15 cleared
  for (int c = tid; c < 16 * 64; c += 256) {
    const int j = c >> 6, ch = c & 63;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (j < Nk) v = *reinterpret_cast<const u32x4*>(a.KV16 + ((size_t)b * Nk + j) * 512 + 8 * ch);
    if (ch < 32) *reinterpret_cast<u32x4*>(Ks + j * PK + 16 * ch) = v;
    else         *reinterpret_cast<u32x4*>(Vs + j * PV + 16 * (ch - 32)) = v;
  }
  // this lane's query fragments (B operand: lane = row, 8 consecutive features)
  const us16* qrow = a.Q16 + (row0 + min(l31, nrows - 1)) * 256 + 64 * w + 8 * h;
  bf16x8 qf[2][2];
#pragma unroll
  for (int hd = 0; hd < 2; ++hd)
#pragma unroll
    for (int s = 0; s < 2; ++s) qf[hd][s] = as_frag(*reinterpret_cast<const u32x4*>(qrow + 32 * hd + 16 * s));
  __syncthreads();
  const bool dodrop = a.drop.p > 0.f;
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
  for (int hd = 0; hd < 2; ++hd) {
    const int head = 2 * w + hd;
    // S^T[j][row] = K_h . Q_h^T  (queries are pre-scaled): lane = row, registers 0..7 = keys acc_row(i, h) < 16
    f32x16 S = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (l31 & 15) * PK + 2 * (32 * head + 16 * s + 8 * h));
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[hd][s], S, 0, 0, 0);
    }
    float e[8];
    rg_softmax(S, h, Nk, e);
    const uint32_t ibase = ((uint32_t)(row0 + l31) * 8u + (uint32_t)head) * (uint32_t)Nk;
    if (dodrop) {
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] *= drop_mult(a.drop, SITE_ATTN_RG2KG, ibase + (uint32_t)acc_row(i, h));
    }
    const bf16x8 pf = as_frag(u32x4{pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])});
    // O^T = V_h^T . P^T: one k step over the 16 keys; A fragment (lane = feature) element jj = key 8 (jj >> 2) + 4 h + (jj & 3)
    const char* vp = Vs + (4 * h + q4) * PV + 2 * (32 * head + 16 * g1 + 4 * p4);
    const bf16x8 vf = join(lds_tr16(vp), lds_tr16(vp + 8 * PV));
    const f32x16 O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, zero16(), 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<u32x2*>(bufO + l31 * PR + 2 * (32 * head + 8 * g + 4 * h)) =
          u32x2{pack2(O[4 * g], O[4 * g + 1]), pack2(O[4 * g + 2], O[4 * g + 3])};
  }
}

// KG->RG attention, one SPLIT: the Nk query rows of sample b against 64 of its RG keys (rows rb + 64 sp ..).  The key ROWS
// sit in the accumulator registers (lane = query j), so the row maximum and the sums are in-lane plus one exchange between
// the lane halves; wave w owns heads 2w, 2w+1; every load of the split is issued up front.  Writes the flash-style partial
// {m[16], l[16], Z[16][32]} per head (relative to the split's own maximum); the last split of a sample to finish combines.
constexpr int SPLIT_ROWS = 64;
constexpr int PART_FLOATS = 16 + 16 + 16 * 32;      // per (sample, split, head)
static_assert(PART_FLOATS == FUSED_PART_FLOATS, "fused_rows.h");
__device__ __forceinline__ void attn_kg_split(const BackArgs& a, char* smem, int b, int sp, int w, int lane) {
  const int l31 = lane & 31, h = lane >> 5, Nk = a.Nk;
  const int rb = a.off[b], nr = a.off[b + 1] - rb, r0 = SPLIT_ROWS * sp;
  char* Vt = smem + w * (32 * PVC);
  const us16* qrow = a.Q2_16 + ((size_t)b * Nk + min(l31, Nk - 1)) * 256 + 64 * w + 8 * h;
  const us16* kbase = a.KV2_16 + (size_t)rb * 512 + 64 * w + 8 * h;
  const us16* vbase = a.KV2_16 + (size_t)rb * 512 + 256 + 64 * w;
  bf16x8 qf[2][2];
  u32x4 kf[2][2][2], vv[2][4];
#pragma unroll
  for (int hd = 0; hd < 2; ++hd)
#pragma unroll
    for (int s = 0; s < 2; ++s) qf[hd][s] = as_frag(*reinterpret_cast<const u32x4*>(qrow + 32 * hd + 16 * s));
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const us16* p = kbase + (size_t)min(r0 + 32 * c + l31, nr - 1) * 512;
#pragma unroll
    for (int hd = 0; hd < 2; ++hd)
#pragma unroll
      for (int s = 0; s < 2; ++s) kf[c][hd][s] = *reinterpret_cast<const u32x4*>(p + 32 * hd + 16 * s);
  }
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // value chunk: 32 rows x 128 bytes (this wave's two heads): slot id = lane + 64 i -> row id >> 3, 16-byte chunk id & 7
      const int id = lane + 64 * i;
      vv[c][i] = *reinterpret_cast<const u32x4*>(vbase + (size_t)min(r0 + 32 * c + (id >> 3), nr - 1) * 512 + 8 * (id & 7));
    }
  // scores S[c][hd][row acc_row(i, h) of chunk c][query l31] (queries are pre-scaled), maximum over the split's valid keys
  f32x16 S[2][2];
  float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int hd = 0; hd < 2; ++hd) {
      S[c][hd] = zero16();
#pragma unroll
      for (int s = 0; s < 2; ++s) S[c][hd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(kf[c][hd][s]), qf[hd][s], S[c][hd], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) mx[hd] = (r0 + 32 * c + acc_row(i, h) < nr) ? fmaxf(mx[hd], S[c][hd][i]) : mx[hd];
    }
  mx[0] = fmaxf(mx[0], __shfl_xor(mx[0], 32, 64));
  mx[1] = fmaxf(mx[1], __shfl_xor(mx[1], 32, 64));
  stamp(a.stamps, 1);
  f32x16 Z[2] = {zero16(), zero16()};
  float L[2] = {0.f, 0.f};
  const bool dodrop = a.drop.p > 0.f;
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int id = lane + 64 * i; *reinterpret_cast<u32x4*>(Vt + (id >> 3) * PVC + 16 * (id & 7)) = vv[c][i]; }
#pragma unroll
    for (int hd = 0; hd < 2; ++hd) {
      float e[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = r0 + 32 * c + acc_row(i, h);
        e[i] = row < nr ? __expf(S[c][hd][i] - mx[hd]) : 0.f;
        L[hd] += e[i];
        if (dodrop) e[i] *= drop_mult(a.drop, SITE_ATTN_KG2RG, ((uint32_t)(rb + row) * 8u + (uint32_t)(2 * w + hd)) * (uint32_t)Nk + (uint32_t)l31);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 ef = as_frag(u32x4{pack2(e[8 * s], e[8 * s + 1]), pack2(e[8 * s + 2], e[8 * s + 3]),
                                        pack2(e[8 * s + 4], e[8 * s + 5]), pack2(e[8 * s + 6], e[8 * s + 7])});
        const char* vp = Vt + (16 * s + 4 * h + q4) * PVC + 2 * (32 * hd + 16 * g1 + 4 * p4);
        const bf16x8 vf = join(lds_tr16(vp), lds_tr16(vp + 8 * PVC));
        Z[hd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ef, vf, Z[hd], 0, 0, 0);      // Z[query][feature] += E^T . V
      }
    }
  }
  L[0] += __shfl_xor(L[0], 32, 64);
  L[1] += __shfl_xor(L[1], 32, 64);
  stamp(a.stamps, 2);
  // partial of this split: Z has the query on its registers (0..7 -> j = acc_row(i, h) < 16) and the feature on the lane
#pragma unroll
  for (int hd = 0; hd < 2; ++hd) {
    // write-through (sc1) stores: the hand-off below then needs no agent-scope release, i.e. no write-back of this XCD's L2
    float* part = a.part + (((size_t)a.tile_off[b] + 2 * sp) * 8 + 2 * w + hd) * PART_FLOATS;       // (a split = two 32-row tiles)
    if (lane < 16) {
      __hip_atomic_store(part + lane, mx[hd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(part + 16 + lane, L[hd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // Z [16 queries][32 features] goes through this wave's (now idle) value-tile scratch so that it leaves as 16-byte
    // write-through stores: a dword sc1 store is one fabric write per lane, six times the time per byte
    float* zt = reinterpret_cast<float*>(Vt);
#pragma unroll
    for (int i = 0; i < 8; ++i) zt[acc_row(i, h) * 32 + l31] = Z[hd][i];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int idx = lane + 64 * r;                       // float4 idx of the [16][32] tile
      const f32x4 v = *reinterpret_cast<const f32x4*>(zt + 4 * idx);
      float* dst = part + 32 + 4 * idx;
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    }
  }
}

// The last split of sample b: combine the partials into the attention output (bf16, rows j < Nk of bufO).  The partials
// were written by other CUs a moment ago, so every read is a long-latency miss: all loops run in batches of independent,
// unconditional loads (split index clamped, contribution masked) instead of one round trip per split.
__device__ __forceinline__ void attn_kg_combine(const BackArgs& a, char* smem, char* bufO, int b, int nsplit) {
  const int tid = threadIdx.x, Nk = a.Nk;
  float* sc = reinterpret_cast<float*>(smem);              // [nsplit][128] scale factors exp(m_s - M), then [128] 1 / L
  float* invL = sc + nsplit * 128;
  const float* part = a.part + (size_t)a.tile_off[b] * 8 * PART_FLOATS;       // split s at + 2 s tiles
  constexpr size_t SS = (size_t)2 * 8 * PART_FLOATS;                          // floats between consecutive splits
  // the first batch of Z loads does not depend on the scale factors: it is issued together with the m / l loads, so the
  // whole combine is (1 + number of further 4-split batches) memory round trips
  // a thread owns four (head, query, 4 features) items: item it = tid + 256 it -> head = it_idx >> 7, query = (it_idx >> 3) & 15,
  // features 4 (it_idx & 7) .. + 3, i.e. one 16-byte load per split and item
  f32x4 zv[4][4];
  auto load_z = [&](int s0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float* z = part + (size_t)min(s0 + k, nsplit - 1) * SS + 32;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = tid + 256 * it;
        zv[k][it] = *reinterpret_cast<const f32x4*>(z + (size_t)(idx >> 7) * PART_FLOATS + 4 * (idx & 127));
      }
    }
  };
  load_z(0);
  if (tid < 128) {
    const int hd = tid >> 4, j = tid & 15;
    const float* p0 = part + (size_t)hd * PART_FLOATS + j;
    float M = -INFINITY, L = 0.f;
    if (nsplit <= 16) {                                      // (block-uniform) the usual case: every m / l value in one batch
      float mv[16], lv[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) { const size_t o = (size_t)min(k, nsplit - 1) * SS; mv[k] = p0[o]; lv[k] = p0[o + 16]; }
#pragma unroll
      for (int k = 0; k < 16; ++k) M = fmaxf(M, mv[k]);
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (k < nsplit) { const float e = __expf(mv[k] - M); sc[k * 128 + tid] = e; L = fmaf(lv[k], e, L); }
    } else {
      for (int s0 = 0; s0 < nsplit; s0 += 8) {
        float mv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) mv[k] = p0[(size_t)min(s0 + k, nsplit - 1) * SS];
#pragma unroll
        for (int k = 0; k < 8; ++k) M = fmaxf(M, mv[k]);
      }
      for (int s0 = 0; s0 < nsplit; s0 += 8) {
        float mv[8], lv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const size_t o = (size_t)min(s0 + k, nsplit - 1) * SS; mv[k] = p0[o]; lv[k] = p0[o + 16]; }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (s0 + k < nsplit) { const float e = __expf(mv[k] - M); sc[(s0 + k) * 128 + tid] = e; L = fmaf(lv[k], e, L); }
      }
    }
    invL[tid] = 1.0f / L;
    if (a.save && a.lse2 && j < Nk) { float* o = a.lse2 + (((size_t)b * 8 + hd) * 16 + j) * 2; o[0] = M; o[1] = L; }
  }
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) acc[it] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < nsplit; s0 += 4) {
    if (s0 > 0) load_z(s0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (s0 + k < nsplit) {
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[it] += zv[k][it] * sc[(s0 + k) * 128 + ((tid + 256 * it) >> 3)];     // (index = head * 16 + query)
      }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + 256 * it, hd = idx >> 7, j = (idx >> 3) & 15, f4 = idx & 7;
    if (j < Nk) {
      const float il = invL[hd * 16 + j];
      *reinterpret_cast<u32x2*>(bufO + j * PR + 2 * (32 * hd + 4 * f4)) =
          u32x2{pack2(acc[it][0] * il, acc[it][1] * il), pack2(acc[it][2] * il, acc[it][3] * il)};
    }
  }
}

// OCC: blocks per CU the register budget is sized for -- 3 (168 VGPRs, a few dwords of scratch) for launches of more than two
// blocks per CU, where the third co-resident tile hides the others' waits (B = 64: 77 -> 68 us, B = 1024: 1.33 -> 1.16 ms);
// 2 for the small batches whose one tile per CU is the critical path (B = 16: the tighter budget cost 1.7 us).
template <int DEPTH, bool ROT, int OCC>
__global__ __launch_bounds__(256, OCC) void back_kernel(const BackArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nkg = a.B * a.max_splits;
  // Dispatch order.  Small batches: the KG attention splits first -- their sample's chain behind them is the kernel's critical
  // path.  Large batches (more blocks than the 2 x 256 slots): one RG tile per CU first, then the splits, then the other tiles
  // -- hundreds of light, latency-bound split blocks alone on the chip were 10 us (B = 64) to 45 us (B = 256) in which no
  // weight stream ran.  vid = the block's id in the splits-first numbering.
  const int vid = (int)blockIdx.x < a.lead_tiles ? nkg + (int)blockIdx.x : ((int)blockIdx.x < a.lead_tiles + nkg ? (int)blockIdx.x - a.lead_tiles : (int)blockIdx.x);
  const bool kg = vid < nkg;
  const int rot = ROT ? (int)(blockIdx.x >> 3) : 0;
  using L = BackL<OCC>;
  char* bufO = smem + L::O; char* bufY = smem + L::Y;
  float* red = reinterpret_cast<float*>(smem + L::RED);
  float* tile32 = reinterpret_cast<float*>(smem);            // (OCC = 2 only)
  int b; size_t rowg0; int nrows; float inv_n;
  stamp(a.stamps, 0);
  const BackStream& S = a.s[kg ? 1 : 0];
  Stage<2, 16, true, DEPTH> sto;
  Stage<4, 16, false, DEPTH> stf;
  if (!kg) sto.prefetch(reinterpret_cast<const u32x4*>(S.Wo) + (size_t)w * (16 * 2 * 64) + lane, rot);   // (flows during the attention)
  if (kg) {
    b = vid / a.max_splits;
    const int sp = vid - b * a.max_splits;
    const int nsplit = (a.off[b + 1] - a.off[b] + SPLIT_ROWS - 1) / SPLIT_ROWS;
    if (sp >= nsplit) return;
    attn_kg_split(a, smem, b, sp, w, lane);
    stamp(a.stamps, 4);
    // hand the partial over (cdna_hip_programming.md, in-launch split-K reduction, write-through form): every storing wave
    // drains its sc1 stores, the block meets, one lane draws a ticket; the block that draws the last one acquires and combines.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(red);
    if (tid == 0) flag[0] = __hip_atomic_fetch_add(a.tickets + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    stamp(a.stamps, 5);
    if (flag[0] != nsplit - 1) return;                                     // (block-uniform)
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int c = tid; c < 32 * PR / 16; c += 256) reinterpret_cast<u32x4*>(bufO)[c] = u32x4{0u, 0u, 0u, 0u};   // rows >= Nk stay zero
    __syncthreads();
    sto.prefetch(reinterpret_cast<const u32x4*>(S.Wo) + (size_t)w * (16 * 2 * 64) + lane, rot);
    attn_kg_combine(a, smem, bufO, b, nsplit);
    stamp(a.stamps, 6);
    rowg0 = (size_t)b * a.Nk; nrows = a.Nk; inv_n = 1.0f / (float)a.Nk;
  } else {
    // one 16-byte load of the batch descriptor's tile table {sample, first row, rows, 1 / Nr} (a binary search of tile_off here was
    // five dependent global loads: 1.5-2 us before a tile's first useful instruction)
    const int4 td = a.tile_desc[vid - nkg];
    if (td.x < 0) return;
    b = td.x; rowg0 = td.y; nrows = td.z; inv_n = __int_as_float(td.w);
    attn_rg_tile(a, smem, bufO, b, rowg0, nrows, w, lane);
  }
  __syncthreads();
  stamp(a.stamps, 1);

  // ---- out-projection + residual, LayerNorm (lane = row; wave w: features 64 w .. 64 w + 63)
  const size_t rrow = rowg0 + min(l31, nrows - 1);
  float u[32];                                            // out-projection + residual, then (in place) the normalised LayerNorm input
  float rstd_keep = 0.f;
  {
    f32x16 acc[2] = {zero16(), zero16()};
    sto.run(bufO + l31 * PR + 16 * h, acc);
    stf.prefetch(reinterpret_cast<const u32x4*>(S.W1) + (size_t)w * (16 * 4 * 64) + lane, rot);           // (flows during the LayerNorm)
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 64 * w + 32 * t + 8 * g + 4 * h;
        const float4 bv = *reinterpret_cast<const float4*>(S.bo + c0);
        const u32x2 rv = *reinterpret_cast<const u32x2*>(S.R16 + rrow * 256 + c0);
        float* o = u + 16 * t + 4 * g;
        o[0] = acc[t][4 * g] + bv.x + bf_lo(rv.x); o[1] = acc[t][4 * g + 1] + bv.y + bf_hi(rv.x);
        o[2] = acc[t][4 * g + 2] + bv.z + bf_lo(rv.y); o[3] = acc[t][4 * g + 3] + bv.w + bf_hi(rv.y);
        part += (o[0] + o[1]) + (o[2] + o[3]);
      }
    // row totals: the lane's 32 values + the other lane half = this wave's 64 features; then the 4 waves through LDS
    auto row_total = [&](float p, int slot) {
      p += __shfl_xor(p, 32, 64);
      if (h == 0) red[slot * 128 + w * 32 + l31] = p;
      __syncthreads();
      return (red[slot * 128 + l31] + red[slot * 128 + 32 + l31]) + (red[slot * 128 + 64 + l31] + red[slot * 128 + 96 + l31]);
    };
    const float mean = row_total(part, 0) * (1.0f / 256.0f);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { u[i] -= mean; sq = fmaf(u[i], u[i], sq); }
    const float rstd = 1.0f / sqrtf(row_total(sq, 1) * (1.0f / 256.0f) + 1e-5f);
    rstd_keep = rstd;
    // (both barriers of row_total are behind every wave's out-projection MFMAs: region A is free for the LayerNorm output tile)
    const bool rok = l31 < nrows;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float ys[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 64 * w + 32 * t + 8 * g + 4 * h;
        const float4 gm = *reinterpret_cast<const float4*>(S.ln_g + c0), bt = *reinterpret_cast<const float4*>(S.ln_b + c0);
        float* o = u + 16 * t + 4 * g;
        const float x0 = o[0] * rstd, x1 = o[1] * rstd, x2 = o[2] * rstd, x3 = o[3] * rstd;
        const float4 y = make_float4(x0 * gm.x + bt.x, x1 * gm.y + bt.y, x2 * gm.z + bt.z, x3 * gm.w + bt.w);
        *reinterpret_cast<u32x2*>(bufY + l31 * PR + 2 * c0) = u32x2{pack2(y.x, y.y), pack2(y.z, y.w)};
        if constexpr (OCC == 3) {
          ys[4 * g] = rok ? y.x : 0.f; ys[4 * g + 1] = rok ? y.y : 0.f; ys[4 * g + 2] = rok ? y.z : 0.f; ys[4 * g + 3] = rok ? y.w : 0.f;
          // saved for the backward: the normalised LayerNorm input, parked in its own tile (leaves as whole rows at the end of the kernel)
          if (a.save) *reinterpret_cast<u32x2*>(smem + L::XH + l31 * PR + 2 * c0) = u32x2{pack2(x0, x1), pack2(x2, x3)};
        } else {
          *reinterpret_cast<float4*>(tile32 + l31 * PT + c0) = y;
          o[0] = x0; o[1] = x1; o[2] = x2; o[3] = x3;       // (kept for the save at the end of the kernel)
        }
      }
      if constexpr (OCC == 2) continue;
      // mean pool of the LayerNorm output: column sums over the 32 lanes of a half by a halving butterfly -- after the steps
      // 16, 8, 4, 2 a lane holds ONE column (index (l31 >> 1) & 15 of its 16), the last step adds the neighbour's half
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int n = 8 >> st, m = 16 >> st;                  // values kept, lane distance
        const bool up = (l31 & m) != 0;
#pragma unroll
        for (int i = 0; i < n; ++i) {
          const float send = up ? ys[i] : ys[i + n];
          const float keep = up ? ys[i + n] : ys[i];
          ys[i] = keep + __shfl_xor(send, m, 64);
        }
      }
      const float tot = ys[0] + __shfl_xor(ys[0], 1, 64);
      if ((l31 & 1) == 0) {
        const int i16 = (l31 >> 1) & 15;
        atomicAdd(S.Ymean + (size_t)b * 256 + 64 * w + 32 * t + 8 * (i16 >> 2) + 4 * h + (i16 & 3), tot * inv_n);
      }
    }
  }
  __syncthreads();
  stamp(a.stamps, 2);
  if constexpr (OCC == 2) {   // mean pool of the LayerNorm output: thread t owns feature t
    float sum = 0.f;
    for (int r = 0; r < nrows; ++r) sum += tile32[r * PT + tid];
    atomicAdd(S.Ymean + (size_t)b * 256 + tid, sum * inv_n);
  }

  // ---- FFN layer 0 + ReLU + dropout, pooled over the rows (lane = feature; wave w: features 128 w .. 128 w + 127)
  {
    f32x16 acc[4] = {zero16(), zero16(), zero16(), zero16()};
    stf.run(bufY + l31 * PR + 16 * h, acc);
    const bool dodrop = a.drop.p > 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int f = 128 * w + 32 * t + l31;
      const float bias = S.b1[f];
      float colsum = 0.f;
      uint32_t wlo = 0u, whi = 0u;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (8 * (i >> 2) >= nrows) continue;                // (block-uniform) a register group whose rows are all past the tile's end: the
                                                            // KG tile has 13 rows, so half of its epilogue -- on the kernel's critical path -- goes
        const int row = acc_row(i, h);
        float v = fmaxf(acc[t][i] + bias, 0.f);
        if (dodrop) v *= drop_mult(a.drop, S.site_ffn, (uint32_t)(rowg0 + row) * 512u + (uint32_t)f);
        if (row >= nrows) v = 0.f;
        colsum += v;
        if (a.save) {
          const unsigned long long bal = __ballot(v > 0.f);
          SET_LANE(wlo, (uint32_t)bal, i); SET_LANE(whi, (uint32_t)(bal >> 32), i);      // lane i <- the ballot's halves (common.h)
        }
      }
      colsum += __shfl_xor(colsum, 32, 64);
      if (h == 0) atomicAdd(S.Hmean + (size_t)b * 512 + f, colsum * inv_n);
      if (a.save && lane < 16) {                       // lane i holds the words of rows acc_row(i, 0) and acc_row(i, 1), features 32 (4 w + t) ..
        const int ra = acc_row(lane, 0);
        if (ra < nrows) S.mask[(rowg0 + ra) * 16 + 4 * w + t] = wlo;
        if (ra + 4 < nrows) S.mask[(rowg0 + ra + 4) * 16 + 4 * w + t] = whi;
      }
    }
  }
  if (a.save) {
    // saved-for-backward tensors, all at the end (no weight stream left to delay): the normalised LayerNorm input goes
    // through its own tile of region A so that it leaves as whole rows like the other two tiles
    char* bufXH = smem + L::XH;
    if (w == 0 && h == 0 && l31 < nrows) S.rstd[rowg0 + l31] = rstd_keep;
    __syncthreads();                                         // (every wave is done reading bufY; OCC = 3: the XH tile was written before the FFN)
    if constexpr (OCC == 2) {                                // region A is free since the mean-pool pass
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float* o = u + 16 * t + 4 * g;
          *reinterpret_cast<u32x2*>(bufXH + l31 * PR + 2 * (64 * w + 32 * t + 8 * g + 4 * h)) = u32x2{pack2(o[0], o[1]), pack2(o[2], o[3])};
        }
      __syncthreads();
    }
    copy_out<5>(bufXH, PR, 0, S.XH16, 256, rowg0, nrows);
    copy_out<5>(bufO, PR, 0, S.O16, 256, rowg0, nrows);
    copy_out<5>(bufY, PR, 0, S.Y16, 256, rowg0, nrows);
  }
  stamp(a.stamps, 3);
}


// ------------------------------------------------------------------------------------------------ backward, first half
// Per tile (RG: 32 rows of a sample; KG: the Nk rows of a sample), from the pooled gradients the per-sample tail left:
//   dH = mask(H) * d(mean H) / n          (virtual: built from the forward's bit mask while the weights stream; also written
//                                           out as the weight-gradient operand)
//   dY = d(mean Z) / n + dH . W1 ;  dU = LayerNorm_backward(dY) (+ dgamma, dbeta) ;  dO = dU . Wo
// RG tiles go on with the RG->KG attention backward (scores recomputed from the saved queries): dQ per row, dK / dV summed
// over the tile into the sample's rows with fp32 atomics.  KG blocks emit dO2 and the softmax row-dots delta2 = dO2 . O2 that
// the second half needs for the KG->RG direction.
constexpr int W_GTAB = 0, W_BUFDU = 1024, W_KS = W_BUFDU + 32 * PR, W_VS = W_KS + 16 * PK, W_BUFDO = W_VS + 16 * PK,
              W_IMGS = W_BUFDO + 32 * PR, W_RED = W_IMGS + 16384, W_LDS = W_RED + 1024;          // 69120 bytes
static_assert(32 * PR + 16384 == 32 * PT * 4, "the fp32 column-sum tile aliases [bufdO | images]");

template <int DEPTH, bool ROT>
__global__ __launch_bounds__(256, 2) void bwd1_kernel(const Bwd1Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // grid order: RG tiles, then the B KG blocks, then the writer blocks.  B + tiles exceeds the CU count by a few blocks at the
  // benchmark size, and the blocks that have to share a CU start ~2.3 us late in this kernel: they had better be the KG
  // blocks (15 us) than RG tiles (18-22 us), which then ended the kernel
  const bool kg = (int)blockIdx.x >= a.rg_tiles_max && (int)blockIdx.x < a.rg_tiles_max + a.B;
  const int rot = ROT ? (int)(blockIdx.x >> 3) : 0;
  if ((int)blockIdx.x >= a.B + a.rg_tiles_max + a.writer_blocks) {       // one clearing block per range (see Bwd1Args)
    const int zi = (int)blockIdx.x - a.B - a.rg_tiles_max - a.writer_blocks;
    u32x4* z = static_cast<u32x4*>(a.zero_ptr[zi]);
    const unsigned n16 = a.zero_bytes[zi] >> 4;
    for (unsigned i = tid; i < n16; i += 256) z[i] = u32x4{0u, 0u, 0u, 0u};
    return;
  }
  if ((int)blockIdx.x >= a.B + a.rg_tiles_max) {
    // writer blocks: materialise dH = mask * d(mean H) / n as the bf16 weight-gradient operand (the tiles above build the same
    // values on the fly and never store them: a store stream in the middle of their weight stream would stall it).
    // A wave takes WR consecutive rows per pass, lane l covers features 8 l .. 8 l + 7 (mask byte l of the row's 64); the pass is
    // two dependent round trips (row -> sample and mask word, then the sample's gradient row) whatever WR is, and these blocks
    // carry the kernel's 69 KB LDS reservation (two per CU: eight waves), so rows in flight per wave is what sets their rate
    // -- one row per pass and a binary search for the sample left a 23 us train of writer blocks behind the tiles at B = 64
    constexpr int WR = 4;
    const int wb = (int)blockIdx.x - a.B - a.rg_tiles_max;
    const int total_rows = a.rows_rg + a.B * a.Nk;
    const float gscale = a.drop.scale;
    for (int r0 = a.writer_first_row + (wb * 4 + (tid >> 6)) * WR; r0 < total_rows; r0 += 4 * WR * a.writer_blocks) {
      int sb[WR]; uint32_t word[WR]; bool isk[WR]; int row[WR];
#pragma unroll
      for (int k = 0; k < WR; ++k) {
        const int r = min(r0 + k, total_rows - 1);
        isk[k] = r >= a.rows_rg;
        row[k] = isk[k] ? r - a.rows_rg : r;
        sb[k] = isk[k] ? row[k] / a.Nk : a.row_sample[row[k]];
        word[k] = a.s[isk[k] ? 1 : 0].mask[(size_t)row[k] * 16 + (lane >> 2)];
      }
      float4 g0[WR], g1[WR]; float inv[WR];
#pragma unroll
      for (int k = 0; k < WR; ++k) {
        const Bwd1Stream& W = a.s[isk[k] ? 1 : 0];
        const float* gp = W.dHm + (size_t)sb[k] * W.ld_dHm + 8 * lane;
        g0[k] = *reinterpret_cast<const float4*>(gp); g1[k] = *reinterpret_cast<const float4*>(gp + 4);
        inv[k] = isk[k] ? 1.0f / (float)a.Nk : a.inv_nr[sb[k]];
      }
#pragma unroll
      for (int k = 0; k < WR; ++k) {
        if (r0 + k >= total_rows) break;
        const uint32_t bits = word[k] >> (8 * (lane & 3));
        const float gs = inv[k] * gscale;
        auto sel = [&](int j, float g) { return ((bits >> j) & 1u) ? (uint32_t)f2bf(g * gs) : 0u; };
        const u32x4 fr = u32x4{sel(0, g0[k].x) | (sel(1, g0[k].y) << 16), sel(2, g0[k].z) | (sel(3, g0[k].w) << 16),
                               sel(4, g1[k].x) | (sel(5, g1[k].y) << 16), sel(6, g1[k].z) | (sel(7, g1[k].w) << 16)};
        *reinterpret_cast<u32x4*>(a.s[isk[k] ? 1 : 0].dH16 + (size_t)row[k] * 512 + 8 * lane) = fr;     // (write-through here: no gain at B = 16, +4 us at B = 64)
      }
    }
    return;
  }
  int b; size_t rowg0; int nrows; float inv_n;
  if (kg) {
    b = (int)blockIdx.x - a.rg_tiles_max; rowg0 = (size_t)b * a.Nk; nrows = a.Nk; inv_n = 1.0f / (float)a.Nk;
  } else {
    const int4 td = a.tile_desc[blockIdx.x];
    if (td.x < 0) return;
    b = td.x; rowg0 = td.y; nrows = td.z; inv_n = __int_as_float(td.w);
  }
  const Bwd1Stream& S = a.s[kg ? 1 : 0];
  us16* gtab = reinterpret_cast<us16*>(smem + W_GTAB);
  char* bufdU = smem + W_BUFDU; char* bufQ = bufdU;                      // (the query tile moves in once dU is consumed)
  char* Ks = smem + W_KS; char* Vs = smem + W_VS; char* bufdO = smem + W_BUFDO; char* imgs = smem + W_IMGS + w * 4096;
  float* red = reinterpret_cast<float*>(smem + W_RED);
  float* tile32 = reinterpret_cast<float*>(smem + W_BUFDO);
  stamp(a.stamps, 0);
  Stage<2, 32, true, DEPTH, false> st1;                                  // k order fixed: the mask words are indexed statically
  st1.prefetch(reinterpret_cast<const u32x4*>(S.W1T) + (size_t)w * (32 * 2 * 64) + lane, 0);
  const size_t vrow = rowg0 + min(l31, nrows - 1);                       // this lane's row, clamped into the tile
  const bool rok = l31 < nrows;
  // per-sample FFN gradient row -> bf16 table (what a set mask bit selects); this row's 512 mask bits
  {
    const float gs = inv_n * a.drop.scale;
    for (int f = tid; f < 512; f += 256) gtab[f] = f2bf(S.dHm[(size_t)b * S.ld_dHm + f] * gs);
  }
  u32x4 mw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) mw[i] = reinterpret_cast<const u32x4*>(S.mask + vrow * 16)[i];
  u32x4 qreg[4];
  if (!kg) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                        // query tile: slot id -> row id >> 5, 16-byte chunk id & 31
      const int id = tid + 256 * i;
      qreg[i] = *reinterpret_cast<const u32x4*>(a.Q16 + (rowg0 + min(id >> 5, nrows - 1)) * 256 + 8 * (id & 31));
    }
    for (int c = tid; c < 16 * 64; c += 256) {                           // the sample's key | value rows, rows Nk..15 cleared
      const int j = c >> 6, ch = c & 63;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (j < a.Nk) v = *reinterpret_cast<const u32x4*>(a.KV16 + ((size_t)b * a.Nk + j) * 512 + 8 * ch);
      if (ch < 32) *reinterpret_cast<u32x4*>(Ks + j * PK + 16 * ch) = v;
      else         *reinterpret_cast<u32x4*>(Vs + j * PK + 16 * (ch - 32)) = v;
    }
  }
  __syncthreads();
  Stage<2, 16, true, DEPTH> st2;
  f32x16 acc1[2] = {zero16(), zero16()};
  st1.run_f([&](int ks) {
    // features 16 ks + 8 h .. + 7 of this row: mask word ks >> 1, bits 16 (ks & 1) + 8 h ..
    const uint32_t word = mw[ks >> 3][(ks >> 1) & 3];
    const uint32_t bits = (word >> (16 * (ks & 1))) >> (8 * h);
    const u32x4 g = *reinterpret_cast<const u32x4*>(smem + W_GTAB + 2 * (16 * ks + 8 * h));
    u32x4 fr;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      fr[d] = (((bits >> (2 * d)) & 1u) ? (g[d] & 0xFFFFu) : 0u) | (((bits >> (2 * d + 1)) & 1u) ? (g[d] & 0xFFFF0000u) : 0u);
    return as_frag(fr);
  }, acc1);
  st2.prefetch(reinterpret_cast<const u32x4*>(S.WoT) + (size_t)w * (16 * 2 * 64) + lane, rot);
  stamp(a.stamps, 1);
  // ---- LayerNorm backward: g = dy * gamma, du = (g - mean(g) - xhat * mean(g * xhat)) * rstd; dgamma += dy * xhat, dbeta += dy
  {
    float gg[32], xh[32];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 64 * w + 32 * t + 8 * g + 4 * h;
        const float4 dc = *reinterpret_cast<const float4*>(S.dcomb + (size_t)b * S.ld_dcomb + c0);
        const float4 gm = *reinterpret_cast<const float4*>(S.ln_g + c0);
        const u32x2 xv = *reinterpret_cast<const u32x2*>(S.XH16 + vrow * 256 + c0);
        float* G = gg + 16 * t + 4 * g; float* X = xh + 16 * t + 4 * g;
        X[0] = bf_lo(xv.x); X[1] = bf_hi(xv.x); X[2] = bf_lo(xv.y); X[3] = bf_hi(xv.y);
        const float d0 = fmaf(dc.x, inv_n, acc1[t][4 * g]), d1 = fmaf(dc.y, inv_n, acc1[t][4 * g + 1]);
        const float d2 = fmaf(dc.z, inv_n, acc1[t][4 * g + 2]), d3 = fmaf(dc.w, inv_n, acc1[t][4 * g + 3]);
        *reinterpret_cast<float4*>(tile32 + l31 * PT + c0) = make_float4(d0 * X[0], d1 * X[1], d2 * X[2], d3 * X[3]);
        acc1[t][4 * g] = d0; acc1[t][4 * g + 1] = d1; acc1[t][4 * g + 2] = d2; acc1[t][4 * g + 3] = d3;      // keep dy
        G[0] = d0 * gm.x; G[1] = d1 * gm.y; G[2] = d2 * gm.z; G[3] = d3 * gm.w;
#pragma unroll
        for (int i = 0; i < 4; ++i) { s1 += G[i]; s2 = fmaf(G[i], X[i], s2); }
      }
    auto row_total = [&](float p, int slot) {
      p += __shfl_xor(p, 32, 64);
      if (h == 0) red[slot * 128 + w * 32 + l31] = p;
      __syncthreads();
      return (red[slot * 128 + l31] + red[slot * 128 + 32 + l31]) + (red[slot * 128 + 64 + l31] + red[slot * 128 + 96 + l31]);
    };
    const float m1 = row_total(s1, 0) * (1.0f / 256.0f);
    const float m2 = row_total(s2, 1) * (1.0f / 256.0f);
    {   // dgamma: column sums of dy * xhat over the tile's rows (thread t owns feature t)
      float sum = 0.f;
      for (int r = 0; r < nrows; ++r) sum += tile32[r * PT + tid];
      atomicAdd(S.dgamma + tid, sum);
    }
    const float rstd = S.rstd[vrow];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 64 * w + 32 * t + 8 * g + 4 * h;
        const float* G = gg + 16 * t + 4 * g; const float* X = xh + 16 * t + 4 * g;
        float du[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) du[i] = rok ? (G[i] - m1 - X[i] * m2) * rstd : 0.f;
        *reinterpret_cast<u32x2*>(bufdU + l31 * PR + 2 * c0) = u32x2{pack2(du[0], du[1]), pack2(du[2], du[3])};
      }
    __syncthreads();                                                     // dgamma pass done with the tile; dU tile complete
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 64 * w + 32 * t + 8 * g + 4 * h;
        *reinterpret_cast<float4*>(tile32 + l31 * PT + c0) = make_float4(acc1[t][4 * g], acc1[t][4 * g + 1], acc1[t][4 * g + 2], acc1[t][4 * g + 3]);
      }
    __syncthreads();
    {   // dbeta: column sums of dy
      float sum = 0.f;
      for (int r = 0; r < nrows; ++r) sum += tile32[r * PT + tid];
      atomicAdd(S.dbeta + tid, sum);
    }
  }
  stamp(a.stamps, 2);
  // ---- dO^T = Wo^T-side product: lane = row, wave w: the 64 features of heads 2w, 2w+1
  f32x16 acc2[2] = {zero16(), zero16()};
  st2.run(bufdU + l31 * PR + 16 * h, acc2);
  copy_out<5>(bufdU, PR, 0, S.dU16, 256, rowg0, nrows);         // (behind the last weight stream of the block)
  if (kg) {
    // dO2 (bf16, the second half's MFMA operand) and delta2[head][j] = sum_f dO2[j][f] * O2[j][f]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float dot = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 64 * w + 32 * t + 8 * g + 4 * h;
        const u32x2 ov = *reinterpret_cast<const u32x2*>(a.O2_16 + vrow * 256 + c0);
        dot = fmaf(acc2[t][4 * g], bf_lo(ov.x), dot); dot = fmaf(acc2[t][4 * g + 1], bf_hi(ov.x), dot);
        dot = fmaf(acc2[t][4 * g + 2], bf_lo(ov.y), dot); dot = fmaf(acc2[t][4 * g + 3], bf_hi(ov.y), dot);
        if (rok) *reinterpret_cast<u32x2*>(a.dO2_16 + (rowg0 + l31) * 256 + c0) =
            u32x2{pack2(acc2[t][4 * g], acc2[t][4 * g + 1]), pack2(acc2[t][4 * g + 2], acc2[t][4 * g + 3])};
      }
      dot += __shfl_xor(dot, 32, 64);
      if (h == 0 && rok) a.delta2[((size_t)b * 8 + 2 * w + t) * 16 + l31] = dot;
    }
    stamp(a.stamps, 3);
    return;
  }
  // ---- RG->KG attention backward
  __syncthreads();                                                       // every wave is done with the dU tile and the fp32 tile
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int id = tid + 256 * i; *reinterpret_cast<u32x4*>(bufQ + (id >> 5) * PR + 16 * (id & 31)) = qreg[i]; }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {                                        // dO tile (bf16): read back transposed for dV
      const int c0 = 64 * w + 32 * t + 8 * g + 4 * h;
      *reinterpret_cast<u32x2*>(bufdO + l31 * PR + 2 * c0) =
          rok ? u32x2{pack2(acc2[t][4 * g], acc2[t][4 * g + 1]), pack2(acc2[t][4 * g + 2], acc2[t][4 * g + 3])} : u32x2{0u, 0u};
    }
  __syncthreads();
  const bool dodrop = a.drop.p > 0.f;
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int head = 2 * w + t;
    // scores and probabilities, exactly as the forward computed them
    f32x16 Sc = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (l31 & 15) * PK + 2 * (32 * head + 16 * s + 8 * h));
      const bf16x8 qf = *reinterpret_cast<const bf16x8*>(bufQ + l31 * PR + 2 * (32 * head + 16 * s + 8 * h));
      Sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf, Sc, 0, 0, 0);
    }
    float pr[8], mm[8];
    rg_softmax(Sc, h, a.Nk, pr);
    const uint32_t ibase = ((uint32_t)(rowg0 + l31) * 8u + (uint32_t)head) * (uint32_t)a.Nk;
#pragma unroll
    for (int i = 0; i < 8; ++i) mm[i] = dodrop ? drop_mult(a.drop, SITE_ATTN_RG2KG, ibase + (uint32_t)acc_row(i, h)) : 1.0f;
    // dPd^T[j][row] = V_h[j] . dO_h[row]: the dO accumulator is the B operand (k order of its registers), so the A fragment
    // takes features 16 s + 4 h .. + 3 and 16 s + 8 + 4 h .. + 3 of value row j
    f32x16 dP = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const char* vp = Vs + (l31 & 15) * PK + 2 * (32 * head + 16 * s + 4 * h);
      const u32x2 v0 = *reinterpret_cast<const u32x2*>(vp), v1 = *reinterpret_cast<const u32x2*>(vp + 16);
      const bf16x8 vf = as_frag(u32x4{v0.x, v0.y, v1.x, v1.y});
      const bf16x8 of = as_frag(u32x4{pack2(acc2[t][8 * s], acc2[t][8 * s + 1]), pack2(acc2[t][8 * s + 2], acc2[t][8 * s + 3]),
                                      pack2(acc2[t][8 * s + 4], acc2[t][8 * s + 5]), pack2(acc2[t][8 * s + 6], acc2[t][8 * s + 7])});
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, of, dP, 0, 0, 0);
    }
    float ds[8], pd[8], delta = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ds[i] = dP[i] * mm[i]; delta = fmaf(pr[i], ds[i], delta); pd[i] = pr[i] * mm[i]; }
    delta += __shfl_xor(delta, 32, 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) { ds[i] = rok ? pr[i] * (ds[i] - delta) : 0.f; if (!rok) pd[i] = 0.f; }
    const u32x4 dsf = u32x4{pack2(ds[0], ds[1]), pack2(ds[2], ds[3]), pack2(ds[4], ds[5]), pack2(ds[6], ds[7])};
    const u32x4 pdf = u32x4{pack2(pd[0], pd[1]), pack2(pd[2], pd[3]), pack2(pd[4], pd[5]), pack2(pd[6], pd[7])};
    {   // dQ^T = scale * K_h^T . dS^T: lane = row
      const char* kp = Ks + (4 * h + q4) * PK + 2 * (32 * head + 16 * g1 + 4 * p4);
      const bf16x8 kt = join(lds_tr16(kp), lds_tr16(kp + 8 * PK));
      const f32x16 dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, as_frag(dsf), zero16(), 0, 0, 0);
      if (rok) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<u32x2*>(a.dQKV16 + (rowg0 + l31) * 768 + 32 * head + 8 * g + 4 * h) =
              u32x2{pack2(dq[4 * g] * a.qscale, dq[4 * g + 1] * a.qscale), pack2(dq[4 * g + 2] * a.qscale, dq[4 * g + 3] * a.qscale)};
      }
    }
    // dS / Pd images [row][16 keys] (bf16): keys 4 h .. + 3 at byte 8 h, keys 8 + 4 h .. at byte 16 + 8 h
    char* imS = imgs + t * 2048; char* imP = imS + 1024;
    *reinterpret_cast<u32x2*>(imS + l31 * 32 + 8 * h) = u32x2{dsf.x, dsf.y};
    *reinterpret_cast<u32x2*>(imS + l31 * 32 + 16 + 8 * h) = u32x2{dsf.z, dsf.w};
    *reinterpret_cast<u32x2*>(imP + l31 * 32 + 8 * h) = u32x2{pdf.x, pdf.y};
    *reinterpret_cast<u32x2*>(imP + l31 * 32 + 16 + 8 * h) = u32x2{pdf.z, pdf.w};
    // dK_h[j][f] += sum_rows dS[j][row] * Qs[row][f];  dV_h[j][f] += sum_rows Pd[j][row] * dO[row][f]  (lane = f, registers = j)
    f32x16 dK = zero16(), dV = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int r0 = 16 * s + 8 * h + q4;
      const bf16x8 sA = join(lds_tr16(imS + r0 * 32 + 8 * p4), lds_tr16(imS + (r0 + 4) * 32 + 8 * p4));
      const bf16x8 pA = join(lds_tr16(imP + r0 * 32 + 8 * p4), lds_tr16(imP + (r0 + 4) * 32 + 8 * p4));
      const int co = 2 * (32 * head + 16 * g1 + 4 * p4);
      const bf16x8 qB = join(lds_tr16(bufQ + r0 * PR + co), lds_tr16(bufQ + (r0 + 4) * PR + co));
      const bf16x8 oB = join(lds_tr16(bufdO + r0 * PR + co), lds_tr16(bufdO + (r0 + 4) * PR + co));
      dK = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sA, qB, dK, 0, 0, 0);
      dV = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pA, oB, dV, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = acc_row(i, h);
      if (j < a.Nk) {
        float* dst = a.dKV + ((size_t)b * a.Nk + j) * 512 + 32 * head + l31;
        atomicAdd(dst, dK[i]);
        atomicAdd(dst + 256, dV[i]);
      }
    }
  }
  stamp(a.stamps, 3);
}

// ------------------------------------------------------------------------------------------------ backward, second half
// RG tiles: the KG->RG attention backward for the tile's 32 key/value rows (probabilities recomputed from the saved softmax
// max / sum), i.e. dK2, dV2 per row and the tile's contribution to the sample's dQ2 (fp32 atomics); then the input
// gradient of the three in-projections in ONE product, dR = dU + [dQ | dK2 | dV2] . [Wq1; Wk2; Wv2]  (K = 768).
// The last tile of a sample to finish (arrival counter, as in the forward) runs the same product for the sample's KG rows:
// dG = dU2 + [dQ2 | dK | dV] . [Wq2; Wk1; Wv1].
constexpr int X_Q2S = 0, X_DO2S = 16 * PK, X_TABS = 2 * 16 * PK, X_BUFT = X_TABS + 1536, X_IMGS = X_BUFT + 32 * PQ,
              X_RED = X_IMGS + 8192, X_LDS = X_RED + 64;                                        // 76352 bytes

template <int DEPTH, bool ROT>
__global__ __launch_bounds__(256, 2) void bwd2_kernel(const Bwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Nk = a.Nk;
  char* Q2s = smem + X_Q2S; char* dO2s = smem + X_DO2S; float* tabs = reinterpret_cast<float*>(smem + X_TABS);
  char* bufT = smem + X_BUFT; char* imgs = smem + X_IMGS;
  int* flag = reinterpret_cast<int*>(smem + X_RED);
  stamp(a.stamps, 0);
  const bool early = (int)blockIdx.x < a.B;          // the B early blocks: the part of dG that needs no RG tile of this launch
  int b;
  if (early) {
    // dGpart = dU2 + [dK | dV] . [Wk1; Wv1]  (k steps 16..47 of the [Wq2; Wk1; Wv1] stream), fp32, and the dK | dV columns of the
    // KG rows' weight-gradient operand.  Counts as one arrival of its sample.
    b = blockIdx.x;
    const size_t krow0 = (size_t)b * Nk;
    Stage<2, 32, true, DEPTH, false> se;
    se.prefetch(reinterpret_cast<const u32x4*>(a.WcKgT) + (size_t)w * (48 * 2 * 64) + 64 * (16 * 2) + lane, 0);
    for (int c = tid; c < 32 * 64; c += 256) {
      const int j = c >> 6, ch = c & 63;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (j < Nk) {
        const float* src = a.dKV + (krow0 + j) * 512 + 8 * ch;
        const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
        v = u32x4{pack2(x0.x, x0.y), pack2(x0.z, x0.w), pack2(x1.x, x1.y), pack2(x1.z, x1.w)};
      }
      *reinterpret_cast<u32x4*>(bufT + j * PQ + 512 + 16 * ch) = v;
    }
    __syncthreads();
    f32x16 acc[2] = {zero16(), zero16()};
    se.run(bufT + l31 * PQ + 512 + 16 * h, acc);
    copy_out<6>(bufT, PQ, 512, a.dQKVkg16 + 256, 768, krow0, Nk);
    if (l31 < Nk) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = 64 * w + 32 * t + 8 * g + 4 * h;
          const u32x2 uv = *reinterpret_cast<const u32x2*>(a.dU2_16 + (krow0 + l31) * 256 + c0);
          float* dst = a.dGpart + (krow0 + l31) * 256 + c0;                 // (write-through: read by another block of this launch)
          __hip_atomic_store(dst, acc[t][4 * g] + bf_lo(uv.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(dst + 1, acc[t][4 * g + 1] + bf_hi(uv.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(dst + 2, acc[t][4 * g + 2] + bf_lo(uv.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(dst + 3, acc[t][4 * g + 3] + bf_hi(uv.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
  } else {
  const int4 td = a.tile_desc[(int)blockIdx.x - a.B];
  if (td.x < 0) return;
  b = td.x;
  const size_t rowg0 = td.y;
  const int nrows = td.z;
  const bool rok = l31 < nrows;
  const size_t vrow = rowg0 + min(l31, nrows - 1);
  Stage<2, 48, true, DEPTH, false> sr;
  sr.prefetch(reinterpret_cast<const u32x4*>(a.WcRgT) + (size_t)w * (48 * 2 * 64) + lane, 0);
  // per-sample inputs: the Nk pre-scaled queries, the gradient of their attention output, softmax max / 1/sum, row-dots
  for (int c = tid; c < 16 * 32; c += 256) {
    const int j = c >> 5, ch = c & 31;
    u32x4 q = u32x4{0u, 0u, 0u, 0u}, o = q;
    if (j < Nk) {
      q = *reinterpret_cast<const u32x4*>(a.Q2_16 + ((size_t)b * Nk + j) * 256 + 8 * ch);
      o = *reinterpret_cast<const u32x4*>(a.dO2_16 + ((size_t)b * Nk + j) * 256 + 8 * ch);
    }
    *reinterpret_cast<u32x4*>(Q2s + j * PK + 16 * ch) = q;
    *reinterpret_cast<u32x4*>(dO2s + j * PK + 16 * ch) = o;
  }
  if (tid < 128) {
    const int j = tid & 15;
    const size_t o = (size_t)b * 128 + tid;                  // [b][head][16]
    const bool ok = j < Nk;
    tabs[tid] = ok ? a.lse2[2 * o] : 0.f;
    tabs[128 + tid] = ok ? 1.0f / a.lse2[2 * o + 1] : 0.f;
    tabs[256 + tid] = ok ? a.delta2[o] : 0.f;
  }
  u32x4 dqreg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {                              // slot id -> row id >> 5, 16-byte chunk id & 31: key tile into LDS, dQ tile into registers
    const int id = tid + 256 * i;
    const size_t r = rowg0 + min(id >> 5, nrows - 1);
    *reinterpret_cast<u32x4*>(bufT + (id >> 5) * PQ + 16 * (id & 31)) = *reinterpret_cast<const u32x4*>(a.KV2_16 + r * 512 + 8 * (id & 31));
    dqreg[i] = *reinterpret_cast<const u32x4*>(a.dQKV16 + r * 768 + 8 * (id & 31));
  }
  bf16x8 kf[2][2], vf[2][2];
  {
    const us16* kp = a.KV2_16 + vrow * 512 + 64 * w + 8 * h;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        kf[t][s] = as_frag(*reinterpret_cast<const u32x4*>(kp + 32 * t + 16 * s));
        vf[t][s] = as_frag(*reinterpret_cast<const u32x4*>(kp + 256 + 32 * t + 16 * s));
      }
  }
  __syncthreads();
  const bool dodrop = a.drop.p > 0.f;
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int head = 2 * w + t;
    f32x16 S2 = zero16(), dP = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int co = (l31 & 15) * PK + 2 * (32 * head + 16 * s + 8 * h);
      S2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Q2s + co), kf[t][s], S2, 0, 0, 0);     // [query j][row]
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(dO2s + co), vf[t][s], dP, 0, 0, 0);    // dO2[j] . V2[row]
    }
    float ds[8], pd[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = acc_row(i, h);
      const float p = (j < Nk && rok) ? __expf(S2[i] - tabs[head * 16 + j]) * tabs[128 + head * 16 + j] : 0.f;
      const float mm = dodrop ? drop_mult(a.drop, SITE_ATTN_KG2RG, ((uint32_t)(rowg0 + l31) * 8u + (uint32_t)head) * (uint32_t)Nk + (uint32_t)j) : 1.0f;
      ds[i] = p * (dP[i] * mm - tabs[256 + head * 16 + j]);
      pd[i] = p * mm;
    }
    const u32x4 dsf = u32x4{pack2(ds[0], ds[1]), pack2(ds[2], ds[3]), pack2(ds[4], ds[5]), pack2(ds[6], ds[7])};
    const u32x4 pdf = u32x4{pack2(pd[0], pd[1]), pack2(pd[2], pd[3]), pack2(pd[4], pd[5]), pack2(pd[6], pd[7])};
    // dV2^T = dO2_h^T . P2d,  dK2^T = Q2s_h^T . dS2: lane = row, registers = the head's 32 features -> bf16 tile [dQ | dK2 | dV2]
    const int tro = (4 * h + q4) * PK + 2 * (32 * head + 16 * g1 + 4 * p4);
    const f32x16 dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join(lds_tr16(dO2s + tro), lds_tr16(dO2s + tro + 8 * PK)), as_frag(pdf), zero16(), 0, 0, 0);
    const f32x16 dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join(lds_tr16(Q2s + tro), lds_tr16(Q2s + tro + 8 * PK)), as_frag(dsf), zero16(), 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * head + 8 * g + 4 * h;
      *reinterpret_cast<u32x2*>(bufT + l31 * PQ + 2 * (256 + c)) = u32x2{pack2(dk[4 * g], dk[4 * g + 1]), pack2(dk[4 * g + 2], dk[4 * g + 3])};
      *reinterpret_cast<u32x2*>(bufT + l31 * PQ + 2 * (512 + c)) = u32x2{pack2(dv[4 * g], dv[4 * g + 1]), pack2(dv[4 * g + 2], dv[4 * g + 3])};
    }
    // dQ2_h[j][f] += scale * sum_rows dS2[j][row] * K2[row][f]: dS2 image [row][16 queries] and the key tile read transposed
    char* im = imgs + head * 1024;
    *reinterpret_cast<u32x2*>(im + l31 * 32 + 8 * h) = u32x2{dsf.x, dsf.y};
    *reinterpret_cast<u32x2*>(im + l31 * 32 + 16 + 8 * h) = u32x2{dsf.z, dsf.w};
    f32x16 dq2 = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int r0 = 16 * s + 8 * h + q4;
      const bf16x8 sA = join(lds_tr16(im + r0 * 32 + 8 * p4), lds_tr16(im + (r0 + 4) * 32 + 8 * p4));
      const int co = 2 * (32 * head + 16 * g1 + 4 * p4);
      const bf16x8 kB = join(lds_tr16(bufT + r0 * PQ + co), lds_tr16(bufT + (r0 + 4) * PQ + co));
      dq2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sA, kB, dq2, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = acc_row(i, h);
      if (j < Nk) atomicAdd(a.dQ2acc + ((size_t)b * Nk + j) * 256 + 32 * head + l31, dq2[i] * a.qscale);
    }
  }
  __syncthreads();                                           // key tile consumed by every wave; dK2 | dV2 tile complete
  stamp(a.stamps, 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int id = tid + 256 * i; *reinterpret_cast<u32x4*>(bufT + (id >> 5) * PQ + 16 * (id & 31)) = dqreg[i]; }
  __syncthreads();
  {
    f32x16 acc[2] = {zero16(), zero16()};
    sr.run(bufT + l31 * PQ + 16 * h, acc);
    copy_out<6>(bufT, PQ, 512, a.dQKV16 + 256, 768, rowg0, nrows);     // (behind the weight stream)
    if (rok) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = 64 * w + 32 * t + 8 * g + 4 * h;
          const u32x2 uv = *reinterpret_cast<const u32x2*>(a.dU16 + (rowg0 + l31) * 256 + c0);
          *reinterpret_cast<u32x2*>(a.dR16 + (rowg0 + l31) * 256 + c0) =
              u32x2{pack2(acc[t][4 * g] + bf_lo(uv.x), acc[t][4 * g + 1] + bf_hi(uv.x)), pack2(acc[t][4 * g + 2] + bf_lo(uv.y), acc[t][4 * g + 3] + bf_hi(uv.y))};
        }
    }
  }
  stamp(a.stamps, 2);
  }
  // ---- arrival: the last block of the sample (its RG tiles and its early block) finishes the KG rows' product.  What it reads from the other
  // tiles are the dQ2 sums, fp32 atomics that execute at the memory side (no L2 line to write back), so a tile only has
  // to have its atomics acknowledged (vmcnt) before its ticket; everything else it reads is the previous launch's.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) flag[0] = __hip_atomic_fetch_add(a.tickets + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (flag[0] != a.tile_off[b + 1] - a.tile_off[b]) return;          // (tiles + the early block)
  Stage<2, 16, true, DEPTH, false> sg;
  sg.prefetch(reinterpret_cast<const u32x4*>(a.WcKgT) + (size_t)w * (48 * 2 * 64) + lane, 0);     // k steps 0..15: Wq2
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // dQ2 of the sample's Nk rows: fp32 sums -> bf16 tile (rows >= Nk cleared)
  const size_t krow0 = (size_t)b * Nk;
  for (int c = tid; c < 32 * 32; c += 256) {
    const int j = c >> 5, ch = c & 31;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (j < Nk) {
      const float* src = a.dQ2acc + (krow0 + j) * 256 + 8 * ch;
      const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
      v = u32x4{pack2(x0.x, x0.y), pack2(x0.z, x0.w), pack2(x1.x, x1.y), pack2(x1.z, x1.w)};
    }
    *reinterpret_cast<u32x4*>(bufT + j * PQ + 16 * ch) = v;
  }
  __syncthreads();
  {
    f32x16 acc[2] = {zero16(), zero16()};
    sg.run(bufT + l31 * PQ + 16 * h, acc);
    copy_out<5>(bufT, PQ, 0, a.dQKVkg16, 768, krow0, Nk);
    if (l31 < Nk) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = 64 * w + 32 * t + 8 * g + 4 * h;
          const float4 pv = *reinterpret_cast<const float4*>(a.dGpart + (krow0 + l31) * 256 + c0);
          *reinterpret_cast<u32x2*>(a.dG16 + (krow0 + l31) * 256 + c0) =
              u32x2{pack2(acc[t][4 * g] + pv.x, acc[t][4 * g + 1] + pv.y), pack2(acc[t][4 * g + 2] + pv.z, acc[t][4 * g + 3] + pv.w)};
        }
    }
  }
  stamp(a.stamps, 3);
}

// ------------------------------------------------------------------------------------------------ backward, second half without the input-gradient products (Bwd2Args::param_space)
// RG tiles: the KG->RG attention backward for the tile's 32 key/value rows (probabilities recomputed from the saved softmax
// max / sum), i.e. dK2, dV2 per row -- the last columns of the weight-gradient operand [dQ | dK2 | dV2] -- and the tile's
// contribution to the sample's dQ2 (fp32 atomics).  The last tile of a sample to finish (arrival counter, as in the forward) turns
// the sample's dQ2 sums into the KG rows' operand columns; B early blocks do the same for the first half's dK | dV sums.
// There is no input-gradient product here any more: nothing needs dR = dU + [dQ|dK2|dV2].W once the projection's weight gradient
// is taken in parameter space (fusion_abi.hip, backward_nodes17):  dW_p = dU^T x + W_in^T (dQKV^T x),  dW_in = (dQKV^T x) W_p^T + db b_p^T.

template <int DEPTH, bool ROT>
__global__ __launch_bounds__(256, 2) void bwd2p_kernel(const Bwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Nk = a.Nk;
  char* Q2s = smem + X_Q2S; char* dO2s = smem + X_DO2S; float* tabs = reinterpret_cast<float*>(smem + X_TABS);
  char* bufT = smem + X_BUFT; char* imgs = smem + X_IMGS;
  int* flag = reinterpret_cast<int*>(smem + X_RED);
  stamp(a.stamps, 0);
  const bool early = (int)blockIdx.x < a.B;          // the B early blocks: the part of dG that needs no RG tile of this launch
  int b;
  if (early) {
    // the dK | dV columns of the KG rows' weight-gradient operand: fp32 sums of the first half -> bf16.  Counts as one arrival of its sample.
    b = blockIdx.x;
    const size_t krow0 = (size_t)b * Nk;
    for (int c = tid; c < 32 * 64; c += 256) {
      const int j = c >> 6, ch = c & 63;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (j < Nk) {
        const float* src = a.dKV + (krow0 + j) * 512 + 8 * ch;
        const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
        v = u32x4{pack2(x0.x, x0.y), pack2(x0.z, x0.w), pack2(x1.x, x1.y), pack2(x1.z, x1.w)};
      }
      *reinterpret_cast<u32x4*>(bufT + j * PQ + 512 + 16 * ch) = v;
    }
    __syncthreads();
    copy_out<6>(bufT, PQ, 512, a.dQKVkg16 + 256, 768, krow0, Nk);
  } else {
  const int4 td = a.tile_desc[(int)blockIdx.x - a.B];
  if (td.x < 0) return;
  b = td.x;
  const size_t rowg0 = td.y;
  const int nrows = td.z;
  const bool rok = l31 < nrows;
  const size_t vrow = rowg0 + min(l31, nrows - 1);
  // per-sample inputs: the Nk pre-scaled queries, the gradient of their attention output, softmax max / 1/sum, row-dots
  for (int c = tid; c < 16 * 32; c += 256) {
    const int j = c >> 5, ch = c & 31;
    u32x4 q = u32x4{0u, 0u, 0u, 0u}, o = q;
    if (j < Nk) {
      q = *reinterpret_cast<const u32x4*>(a.Q2_16 + ((size_t)b * Nk + j) * 256 + 8 * ch);
      o = *reinterpret_cast<const u32x4*>(a.dO2_16 + ((size_t)b * Nk + j) * 256 + 8 * ch);
    }
    *reinterpret_cast<u32x4*>(Q2s + j * PK + 16 * ch) = q;
    *reinterpret_cast<u32x4*>(dO2s + j * PK + 16 * ch) = o;
  }
  if (tid < 128) {
    const int j = tid & 15;
    const size_t o = (size_t)b * 128 + tid;                  // [b][head][16]
    const bool ok = j < Nk;
    tabs[tid] = ok ? a.lse2[2 * o] : 0.f;
    tabs[128 + tid] = ok ? 1.0f / a.lse2[2 * o + 1] : 0.f;
    tabs[256 + tid] = ok ? a.delta2[o] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {                              // slot id -> row id >> 5, 16-byte chunk id & 31: key tile into LDS
    const int id = tid + 256 * i;
    const size_t r = rowg0 + min(id >> 5, nrows - 1);
    *reinterpret_cast<u32x4*>(bufT + (id >> 5) * PQ + 16 * (id & 31)) = *reinterpret_cast<const u32x4*>(a.KV2_16 + r * 512 + 8 * (id & 31));
  }
  bf16x8 kf[2][2], vf[2][2];
  {
    const us16* kp = a.KV2_16 + vrow * 512 + 64 * w + 8 * h;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        kf[t][s] = as_frag(*reinterpret_cast<const u32x4*>(kp + 32 * t + 16 * s));
        vf[t][s] = as_frag(*reinterpret_cast<const u32x4*>(kp + 256 + 32 * t + 16 * s));
      }
  }
  __syncthreads();
  const bool dodrop = a.drop.p > 0.f;
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int head = 2 * w + t;
    f32x16 S2 = zero16(), dP = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int co = (l31 & 15) * PK + 2 * (32 * head + 16 * s + 8 * h);
      S2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Q2s + co), kf[t][s], S2, 0, 0, 0);     // [query j][row]
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(dO2s + co), vf[t][s], dP, 0, 0, 0);    // dO2[j] . V2[row]
    }
    float ds[8], pd[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = acc_row(i, h);
      const float p = (j < Nk && rok) ? __expf(S2[i] - tabs[head * 16 + j]) * tabs[128 + head * 16 + j] : 0.f;
      const float mm = dodrop ? drop_mult(a.drop, SITE_ATTN_KG2RG, ((uint32_t)(rowg0 + l31) * 8u + (uint32_t)head) * (uint32_t)Nk + (uint32_t)j) : 1.0f;
      ds[i] = p * (dP[i] * mm - tabs[256 + head * 16 + j]);
      pd[i] = p * mm;
    }
    const u32x4 dsf = u32x4{pack2(ds[0], ds[1]), pack2(ds[2], ds[3]), pack2(ds[4], ds[5]), pack2(ds[6], ds[7])};
    const u32x4 pdf = u32x4{pack2(pd[0], pd[1]), pack2(pd[2], pd[3]), pack2(pd[4], pd[5]), pack2(pd[6], pd[7])};
    // dV2^T = dO2_h^T . P2d,  dK2^T = Q2s_h^T . dS2: lane = row, registers = the head's 32 features -> bf16 tile [dQ | dK2 | dV2]
    const int tro = (4 * h + q4) * PK + 2 * (32 * head + 16 * g1 + 4 * p4);
    const f32x16 dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join(lds_tr16(dO2s + tro), lds_tr16(dO2s + tro + 8 * PK)), as_frag(pdf), zero16(), 0, 0, 0);
    const f32x16 dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join(lds_tr16(Q2s + tro), lds_tr16(Q2s + tro + 8 * PK)), as_frag(dsf), zero16(), 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * head + 8 * g + 4 * h;
      *reinterpret_cast<u32x2*>(bufT + l31 * PQ + 2 * (256 + c)) = u32x2{pack2(dk[4 * g], dk[4 * g + 1]), pack2(dk[4 * g + 2], dk[4 * g + 3])};
      *reinterpret_cast<u32x2*>(bufT + l31 * PQ + 2 * (512 + c)) = u32x2{pack2(dv[4 * g], dv[4 * g + 1]), pack2(dv[4 * g + 2], dv[4 * g + 3])};
    }
    // dQ2_h[j][f] += scale * sum_rows dS2[j][row] * K2[row][f]: dS2 image [row][16 queries] and the key tile read transposed
    char* im = imgs + head * 1024;
    *reinterpret_cast<u32x2*>(im + l31 * 32 + 8 * h) = u32x2{dsf.x, dsf.y};
    *reinterpret_cast<u32x2*>(im + l31 * 32 + 16 + 8 * h) = u32x2{dsf.z, dsf.w};
    f32x16 dq2 = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int r0 = 16 * s + 8 * h + q4;
      const bf16x8 sA = join(lds_tr16(im + r0 * 32 + 8 * p4), lds_tr16(im + (r0 + 4) * 32 + 8 * p4));
      const int co = 2 * (32 * head + 16 * g1 + 4 * p4);
      const bf16x8 kB = join(lds_tr16(bufT + r0 * PQ + co), lds_tr16(bufT + (r0 + 4) * PQ + co));
      dq2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sA, kB, dq2, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = acc_row(i, h);
      if (j < Nk) atomicAdd(a.dQ2acc + ((size_t)b * Nk + j) * 256 + 32 * head + l31, dq2[i] * a.qscale);
    }
  }
  __syncthreads();                                           // key tile consumed by every wave; dK2 | dV2 tile complete
  stamp(a.stamps, 1);
  copy_out<6>(bufT, PQ, 512, a.dQKV16 + 256, 768, rowg0, nrows);     // the weight-gradient operand's dK2 | dV2 columns (dQ: the first half's)
  stamp(a.stamps, 2);
  }
  // large batches: no arrival protocol (a drain of the tile's stores, a ticket round trip and three barriers per 32-row tile: a quarter
  // of a tile block's life) -- bwd2_finish_kernel converts the dQ2 sums behind this launch
  if (a.split_finish) return;
  // ---- arrival: the last block of the sample (its RG tiles and its early block) finishes the KG rows' product.  What it reads from the other
  // tiles are the dQ2 sums, fp32 atomics that execute at the memory side (no L2 line to write back), so a tile only has
  // to have its atomics acknowledged (vmcnt) before its ticket; everything else it reads is the previous launch's.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) flag[0] = __hip_atomic_fetch_add(a.tickets + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (flag[0] != a.tile_off[b + 1] - a.tile_off[b]) return;          // (tiles + the early block)
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // dQ2 of the sample's Nk rows: fp32 sums -> bf16 tile (rows >= Nk cleared)
  const size_t krow0 = (size_t)b * Nk;
  for (int c = tid; c < 32 * 32; c += 256) {
    const int j = c >> 5, ch = c & 31;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (j < Nk) {
      const float* src = a.dQ2acc + (krow0 + j) * 256 + 8 * ch;
      const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
      v = u32x4{pack2(x0.x, x0.y), pack2(x0.z, x0.w), pack2(x1.x, x1.y), pack2(x1.z, x1.w)};
    }
    *reinterpret_cast<u32x4*>(bufT + j * PQ + 16 * ch) = v;
  }
  __syncthreads();
  copy_out<5>(bufT, PQ, 0, a.dQKVkg16, 768, krow0, Nk);
  stamp(a.stamps, 3);
}

// dQ2 of every sample's Nk rows: fp32 sums (complete: the previous launch's atomics) -> columns 0 .. 255 of the KG rows' weight-gradient
// operand (bf16) -- what the last-arriving tile block of bwd2p_kernel does when the arrival protocol is on.  One block per sample.
__global__ __launch_bounds__(256) void bwd2_finish_kernel(const Bwd2Args a) {
  const size_t krow0 = (size_t)blockIdx.x * a.Nk;
  for (int c = threadIdx.x; c < a.Nk * 32; c += 256) {
    const int j = c >> 5, ch = c & 31;
    const float* src = a.dQ2acc + (krow0 + j) * 256 + 8 * ch;
    const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
    *reinterpret_cast<u32x4*>(a.dQKVkg16 + (krow0 + j) * 768 + 8 * ch) = u32x4{pack2(x0.x, x0.y), pack2(x0.z, x0.w), pack2(x1.x, x1.y), pack2(x1.z, x1.w)};
  }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

size_t fused_front_lds() { return F_LDS; }
size_t fused_back_lds() { return B_LDS; }

int launch_weight_shadows(ShadowBatch& sb, hipStream_t stream) {
  if (sb.n < 0 || sb.n > SHADOW_MAXJ) return (int)hipErrorInvalidValue;
  int total = 0;
  for (int i = 0; i < sb.n; ++i) {
    ShadowJob& J = sb.j[i];
    if ((J.N & 127) || (J.K & 15) || J.nsrc < 1 || J.nsrc > 4 || !J.dst || !al16(J.dst)) return (int)hipErrorInvalidValue;
    int sum = 0;
    for (int s = 0; s < J.nsrc; ++s) {
      if (!J.src[s] || (J.rows[s] & 7) || (!J.transposed && (!al16(J.src[s]) || (J.ld[s] & 3)))) return (int)hipErrorInvalidValue;
      sum += J.rows[s];
    }
    if (sum != (J.transposed ? J.K : J.N)) return (int)hipErrorInvalidValue;
    J.chunk_begin = total;
    total += J.N * J.K / 8;
  }
  if (sb.nzero < 0 || sb.nzero > SHADOW_MAXZ) return (int)hipErrorInvalidValue;
  for (int i = 0; i < sb.nzero; ++i)
    if (!sb.zero_ptr[i] || !al16(sb.zero_ptr[i]) || (sb.zero_bytes[i] & 15)) return (int)hipErrorInvalidValue;
  int blocks = (total + 255) / 256;
  if (sb.nzero && blocks < 64) blocks = 64;
  if (blocks == 0) return 0;
  const int prof = gemm_prof_open(stream, 0.0, PROF_SHADOW);
  hipLaunchKernelGGL(shadow_kernel, dim3(blocks), dim3(256), 0, stream, sb, total);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

int launch_fused_front(FrontArgs& a, int variant, hipStream_t stream) {
  int total = 0;
  for (int i = 0; i < 2; ++i) {
    FrontStream& S = a.s[i];
    if (S.M < 1 || !S.X || !S.W0 || !S.W1 || !S.b0 || !S.bq || !S.bkv || !S.R16 || !S.Q16 || !S.KV16 || (a.save && !S.X16))
      return (int)hipErrorInvalidValue;
    if (!al16(S.X) || !al16(S.b0) || !al16(S.bq) || !al16(S.bkv) || !al16(S.R16) || !al16(S.Q16) || !al16(S.KV16) || !al16(S.W0) || !al16(S.W1))
      return (int)hipErrorInvalidValue;
    S.tile_begin = total;
    total += (S.M + 31) / 32;
  }
  if (a.nzero < 0 || a.nzero > FUSED_FRONT_MAXZ) return (int)hipErrorInvalidValue;
  for (int i = 0; i < a.nzero; ++i)
    if (!a.zero_ptr[i] || !al16(a.zero_ptr[i]) || (a.zero_bytes[i] & 15)) return (int)hipErrorInvalidValue;
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&front_kernel<12, false>), hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&front_kernel<12, true>), hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&front_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
    return true;
  }();
  (void)attr;
  // executed FLOPs: per row 128 -> 256 and 256 -> 768
  const int prof = gemm_prof_open(stream, 2.0 * ((double)a.s[0].M + a.s[1].M) * (128.0 * 256.0 + 256.0 * 768.0), PROF_FRONT);
  if (variant == 0)      hipLaunchKernelGGL((front_kernel<12, false>), dim3(total), dim3(256), F_LDS, stream, a);
  else if (variant == 2) hipLaunchKernelGGL((front_kernel<16, true>), dim3(total), dim3(256), F_LDS, stream, a);
  else                   hipLaunchKernelGGL((front_kernel<12, true>), dim3(total), dim3(256), F_LDS, stream, a);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

thread_local int g_back_lead_mode = 1;
int launch_fused_back(BackArgs& a, int variant, hipStream_t stream) {
  if (a.B < 1 || a.Nk < 1 || a.Nk > 16 || a.rg_tiles_max < 1 || !a.Q16 || !a.KV16 || !a.Q2_16 || !a.KV2_16 || !a.off || !a.tile_off || !a.tile_desc || !a.inv_nr ||
      !a.part || !a.tickets || a.max_splits < 1 || a.max_splits > FUSED_MAX_SPLITS)
    return (int)hipErrorInvalidValue;
  for (int i = 0; i < 2; ++i) {
    const BackStream& S = a.s[i];
    if (!S.Wo || !S.bo || !S.W1 || !S.b1 || !S.ln_g || !S.ln_b || !S.R16 || !S.Ymean || !S.Hmean) return (int)hipErrorInvalidValue;
    if (a.save && (!S.O16 || !S.Y16 || !S.XH16 || !S.rstd || !S.mask)) return (int)hipErrorInvalidValue;
    if (!al16(S.bo) || !al16(S.ln_g) || !al16(S.ln_b) || !al16(S.Wo) || !al16(S.W1) || !al16(S.R16)) return (int)hipErrorInvalidValue;
  }
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&back_kernel<12, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, BackL<2>::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&back_kernel<12, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, BackL<2>::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&back_kernel<12, true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, BackL<3>::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&back_kernel<16, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, BackL<2>::LDS);
    return true;
  }();
  (void)attr;
  const dim3 grid(a.B * a.max_splits + a.rg_tiles_max);
  a.lead_tiles = (g_back_lead_mode != 0 && a.rows_rg / 32 >= 256 && (int)grid.x > 512) ? 256 : 0;      // (rows / 32 <= number of real tiles)
  // executed FLOPs per row: out-projection 256 -> 256, FFN layer 0 256 -> 512, both attention directions (2 x 2 x Nk x 256)
  const double rows = (double)a.rows_rg + (double)a.B * a.Nk;
  const int prof = gemm_prof_open(stream, 2.0 * rows * (256.0 * 256.0 + 256.0 * 512.0) + 8.0 * (double)a.rows_rg * a.Nk * 256.0, PROF_BACK);
  if (variant == 0)        hipLaunchKernelGGL((back_kernel<12, false, 2>), grid, dim3(256), BackL<2>::LDS, stream, a);
  else if (variant == 2)   hipLaunchKernelGGL((back_kernel<16, true, 2>), grid, dim3(256), BackL<2>::LDS, stream, a);
  else if (grid.x > 512u)  hipLaunchKernelGGL((back_kernel<12, true, 3>), grid, dim3(256), BackL<3>::LDS, stream, a);
  else                     hipLaunchKernelGGL((back_kernel<12, true, 2>), grid, dim3(256), BackL<2>::LDS, stream, a);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

size_t fused_bwd1_lds() { return W_LDS; }

int launch_fused_bwd1(Bwd1Args& a, int variant, hipStream_t stream, int kg_only) {
  if (a.B < 1 || a.Nk < 1 || a.Nk > 16 || a.rg_tiles_max < 1 || !a.Q16 || !a.KV16 || !a.dQKV16 || !a.dKV || !a.O2_16 || !a.dO2_16 || !a.delta2 ||
      !a.off || !a.tile_off || !a.tile_desc || !a.inv_nr || !a.row_sample)
    return (int)hipErrorInvalidValue;
  for (int i = 0; i < 2; ++i) {
    const Bwd1Stream& S = a.s[i];
    if (!S.W1T || !S.WoT || !S.mask || !S.XH16 || !S.rstd || !S.ln_g || !S.dHm || !S.dcomb || !S.dH16 || !S.dU16 || !S.dgamma || !S.dbeta)
      return (int)hipErrorInvalidValue;
    if (!al16(S.W1T) || !al16(S.WoT) || !al16(S.mask) || !al16(S.ln_g) || !al16(S.dcomb) || (S.ld_dcomb & 3) || !al16(S.dH16) || !al16(S.dU16))
      return (int)hipErrorInvalidValue;
  }
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd1_kernel<12, false>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd1_kernel<12, true>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
    return true;
  }();
  (void)attr;
  // kg_only (launch_wide2_bwd1 takes the RG rows, their dH16 included): no RG tile blocks, writer blocks for the KG stream's rows only
  a.writer_first_row = kg_only ? a.rows_rg : 0;
  a.writer_blocks = (a.rows_rg + a.B * a.Nk - a.writer_first_row + 15) / 16;   // four rows per wave per pass (see the kernel; eight: the same at B = 64, -5 us at B = 256)
  if (a.writer_blocks > 8192) a.writer_blocks = 8192;
  if (a.nzero < 0 || a.nzero > FUSED_BWD1_MAXZ) return (int)hipErrorInvalidValue;
  for (int i = 0; i < a.nzero; ++i)
    if (!a.zero_ptr[i] || !al16(a.zero_ptr[i]) || (a.zero_bytes[i] & 15)) return (int)hipErrorInvalidValue;
  Bwd1Args k = a;                                                    // (the kernel decodes a block's role from rg_tiles_max)
  if (kg_only) { k.rg_tiles_max = 0; k.stamps = nullptr; }           // (the timeline region belongs to the RG rows' launch then)
  const dim3 grid(k.B + k.rg_tiles_max + k.writer_blocks + k.nzero);
  // executed FLOPs per row: dY (512 -> 256), dO (256 -> 256); RG rows: the RG->KG attention backward (5 products of Nk x 256)
  const double rows = (kg_only ? 0.0 : (double)a.rows_rg) + (double)a.B * a.Nk;
  const int prof = gemm_prof_open(stream, 2.0 * rows * (512.0 * 256.0 + 256.0 * 256.0) + (kg_only ? 0.0 : 10.0 * (double)a.rows_rg * a.Nk * 256.0), PROF_BWD1);
  if (variant == 0) hipLaunchKernelGGL((bwd1_kernel<12, false>), grid, dim3(256), W_LDS, stream, k);
  else              hipLaunchKernelGGL((bwd1_kernel<12, true>), grid, dim3(256), W_LDS, stream, k);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

size_t fused_bwd2_lds() { return X_LDS; }

int launch_fused_bwd2(Bwd2Args& a, int variant, hipStream_t stream) {
  (void)variant;
  if (a.B < 1 || a.Nk < 1 || a.Nk > 16 || a.rg_tiles_max < 1 || !a.Q2_16 || !a.dO2_16 || !a.lse2 || !a.delta2 || !a.KV2_16 || !a.dQKV16 ||
      !a.dU16 || !a.WcRgT || !a.dR16 || !a.dQ2acc || !a.dKV || !a.dU2_16 || !a.WcKgT || !a.dQKVkg16 || !a.dG16 || !a.dGpart || !a.tickets || !a.off || !a.tile_off || !a.tile_desc)
    return (int)hipErrorInvalidValue;
  if (!al16(a.dQKV16) || !al16(a.dQKVkg16) || !al16(a.WcRgT) || !al16(a.WcKgT) || !al16(a.dQ2acc) || !al16(a.dKV)) return (int)hipErrorInvalidValue;
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd2_kernel<12, false>), hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd2p_kernel<12, false>), hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS);
    return true;
  }();
  (void)attr;
  // executed FLOPs per row: dR / dG (768 -> 256) unless the caller takes those gradients in parameter space; RG rows: the KG->RG
  // attention backward (5 products of Nk x 256)
  const double rows = (double)a.rows_rg + (double)a.B * a.Nk;
  const int prof = gemm_prof_open(stream, (a.param_space ? 0.0 : 2.0 * rows * 768.0 * 256.0) + 10.0 * (double)a.rows_rg * a.Nk * 256.0, PROF_BWD2);
  if (!a.param_space) a.split_finish = 0;
  if (a.param_space) hipLaunchKernelGGL((bwd2p_kernel<12, false>), dim3(a.B + a.rg_tiles_max), dim3(256), X_LDS, stream, a);
  else               hipLaunchKernelGGL((bwd2_kernel<12, false>), dim3(a.B + a.rg_tiles_max), dim3(256), X_LDS, stream, a);
  if (a.split_finish) hipLaunchKernelGGL(bwd2_finish_kernel, dim3(a.B), dim3(256), 0, stream, a);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}
