// Wide row tiles for large batches (gfx950): the forward kernels of fused_rows.hip re-cut so that one weight fragment feeds
// several MFMAs.  Contract and argument blocks: fused_rows.h (the same FrontArgs / BackArgs, the same weight shadows and the
// same per-32-row tile table).
//
// Why.  A 32-row tile streams every weight of its chain from L2 once (front 459 KB, back 393 KB) for 13-22 MFLOP: the CU is
// busy 20 us per tile for 1.5 us of MFMA work, and that ratio does not improve with the batch size -- a larger batch is just
// more tiles (round 2: the forward sits at 11-12 % of the bf16 MFMA peak from B = 64 up).  Here a block owns RT consecutive
// 32-row tiles of the tile table (RT = 4: 128 rows) and 8 waves: wave w owns 1/8 of a layer's output features for ALL the
// block's rows, loads each of its weight fragments ONCE into registers and issues RT MFMAs with it, one per sub-tile, whose
// other operand comes out of the block's bf16 activation tiles in LDS.  Per row that is 1/RT of the L2 weight stream and of the
// per-tile fixed costs (barriers, prefetch ramps, epilogue latencies); two waves per SIMD cover each other's LDS / epilogue
// phases.  Sub-tiles keep the 32-row tile's identity (sample, first row, rows, 1/Nr from the batch descriptor), so
// everything per sample -- the 13 keys a row attends to, pooled sums, dropout indices, saved tensors -- is exactly what the
// 32-row kernels compute; the two families are interchangeable per launch (tests/test_hip_wide.py runs every stage test on both).
//
// KG side.  The 32-row back kernel ran the KG->RG attention as ceil(Nr / 64) extra "split" blocks per sample.  At large B
// those are hundreds of light blocks that each hold a CU slot.  Here the RG blocks themselves produce the flash-style partials
// {max, sum, Z} of the sample's 13 queries against THEIR OWN rows as keys (a few MFMAs per sub-tile), one partial per run of
// sub-tiles of the same sample; the block that completes a sample (arrival counter) combines the partials and runs the 13-row
// KG chain at the end of its own work.
#include "fused_rows.h"
#include "shadow_inl.h"
#include "gemm.h"      // launch timing hooks

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
// Developer timeline (testing hook "stamps"): lane 0 of every wave records the 100 MHz wall clock at phase boundaries:
// stamps[(block * 8 + wave) * 16 + k].  Product calls pass null and execute none of it.
__device__ __forceinline__ void stamp(unsigned long long* stamps, int k) {
  if (stamps && (threadIdx.x & 63) == 0) {
    unsigned long long* p = stamps + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16;
    p[k] = __builtin_amdgcn_s_memrealtime();
    if (k == 0) p[11] = __builtin_amdgcn_s_memtime();        // shader-clock counter at the block's first and last stamp (slots 11, 15):
    if (k == 12) p[15] = __builtin_amdgcn_s_memtime();       // the clock the CU actually ran at = their difference / the wall-clock interval
  }
}
__device__ __forceinline__ void store16_wt(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");   // (s_nop: the data registers may be rewritten right behind an asm store)
}

constexpr int NW = 8;        // waves per block
constexpr int NTH = 64 * NW;
constexpr int PX = 272;      // row pitch (bytes) of a [rows][128] bf16 tile
constexpr int PR = 528;      // ... of a [rows][256] bf16 tile (a ds_read_b128 lane group's 16 rows land on 16 distinct bank quads)
constexpr int PQ = 1552;     // ... of a [rows][768] bf16 tile
constexpr int PV = 576;      // the 16 value rows of a sample [16][256] (transposing reads: rows 16 banks apart)

// ---- one linear layer over RT sub-tiles: acc[s][t] (+)= sum over KS k steps.  `wp` = this wave's first fragment + lane (16-byte
// units); fragment (ks, t) is wp[64 (ks KST + t)] (KST = tiles per k step in the shadow's 4-wave layout).  frag(s, ks) = this
// lane's activation fragment of sub-tile s, k step ks.  DEPTH weight fragments are kept in flight; the scheduling barrier pins
// {issue the load DEPTH fragments ahead, (first tile of a k step: start the NEXT k step's activation reads), RT MFMAs}.
template <int RT, int NT, int KS, int KST, int DEPTH>
struct StageW {
  static constexpr int TOTAL = NT * KS;
  static constexpr int D = DEPTH < TOTAL ? DEPTH : TOTAL;
  u32x4 buf[D];
  const u32x4* wp;
  __device__ __forceinline__ const u32x4* wptr(int i) const { return wp + 64 * ((i / NT) * KST + (i % NT)); }
  __device__ __forceinline__ void prefetch(const u32x4* __restrict__ wp_) {
    wp = wp_;
#pragma unroll
    for (int i = 0; i < D; ++i) buf[i] = *wptr(i);
  }
  // ZERO: acc = the products alone (the first k step's MFMAs take the constant 0 as their addend: no accumulator clears)
  template <bool W_IS_A, bool ZERO = false, class F>
  __device__ __forceinline__ void run_f(F&& frag, f32x16 (&acc)[RT][NT]) {
    bf16x8 x[RT], xn[RT];
#pragma unroll
    for (int s = 0; s < RT; ++s) xn[s] = frag(s, 0);
#pragma unroll
    for (int i = 0; i < TOTAL; ++i) {
      const int ks = i / NT, t = i % NT;
      const bf16x8 wf = as_frag(buf[i % D]);
      if (i + D < TOTAL) buf[i % D] = *wptr(i + D);
      if (t == 0) {
#pragma unroll
        for (int s = 0; s < RT; ++s) x[s] = xn[s];
        if (ks + 1 < KS) {
#pragma unroll
          for (int s = 0; s < RT; ++s) xn[s] = frag(s, ks + 1);
        }
        // the NEXT k step's activation reads go out in FRONT of this step's MFMAs (left alone, hipcc sinks them behind the
        // MFMAs to save registers, and every k step then waits out an LDS round trip)
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < RT; ++s) {
        const f32x16 c = (ZERO && ks == 0) ? zero16() : acc[s][t];
        if constexpr (W_IS_A) acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, x[s], c, 0, 0, 0);
        else                  acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[s], wf, c, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // activation tiles in LDS: `act` = address of this lane's first fragment of sub-tile 0 (row lane & 31, byte 16 (lane >> 5)),
  // sub-tile s is `sub` bytes further, k step ks 32 bytes further
  template <bool W_IS_A, bool ZERO = false>
  __device__ __forceinline__ void run(const char* act, int sub, f32x16 (&acc)[RT][NT]) {
    run_f<W_IS_A, ZERO>([&](int s, int ks) { return *reinterpret_cast<const bf16x8*>(act + s * sub + 32 * ks); }, acc);
  }
};

template <int RT, int NT>
__device__ __forceinline__ void clear_acc(f32x16 (&acc)[RT][NT]) {
#pragma unroll
  for (int s = 0; s < RT; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[s][t] = zero16();
}

// rows [0, nrows) of a bf16 LDS tile -> global, 16 bytes per thread, whole rows (LOG2C: log2 of 16-byte chunks per row)
template <int LOG2C, bool WT>
__device__ __forceinline__ void copy_out(const char* lds, int pitch, int col_byte0, us16* dst, int ld, size_t row0, int nrows, int maxrows) {
  for (int c = threadIdx.x; c < (maxrows << LOG2C); c += NTH) {
    const int r = c >> LOG2C, k = c & ((1 << LOG2C) - 1);
    if (r < nrows) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(lds + r * pitch + col_byte0 + 16 * k);
      void* p = reinterpret_cast<char*>(dst + (row0 + r) * (size_t)ld) + 16 * k;
      if constexpr (WT) store16_wt(p, v); else *reinterpret_cast<u32x4*>(p) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ forward, front half
// x -> bf16 -> R = x W0^T + b0 (128 -> 256) -> [q / sqrt(32) | k' | v'] = R W1^T + b (256 -> 768), 32 RT packed rows per block.
// The 768-wide output leaves in three passes of one 32-feature tile per wave.  Each wave packs its [32 RT rows][32 features]
// result through a PRIVATE staging strip and stores it itself (64-byte row segments, 16 bytes per lane), so the pass loop has no
// block barrier at all: the two waves of a SIMD drift apart and one's epilogue + stores run under the other's MFMAs.  Biases
// sit in LDS from the start of the kernel (a global load in an epilogue is a 1-2 us stall of the whole wave, once per pass).
constexpr int PS = 80;       // row pitch (bytes) of a wave's staging strip [rows][32 features]
template <int RT> struct FrontCfg {
  static constexpr int BUFR = 0, BUFS = 32 * RT * PR;
  static constexpr int STG = NW * 32 * RT * PS, XT = 32 * RT * PX;       // staging strips (the bf16 input tile lives there first)
  static constexpr int CONSTS = BUFS + (STG > XT ? STG : XT), LDS = CONSTS + 1024 * 4;      // b0 [256] | bq [256] | bkv [512]
};

template <int RT, int DEPTH>
__global__ __launch_bounds__(NTH, 2) void front8_kernel(const FrontArgs a) {
  using Cfg = FrontCfg<RT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x >= a.xblock0) {                           // extra blocks: weight-shadow chunks (FrontArgs::xjob)
    const int gid = ((int)blockIdx.x - a.xblock0) * NTH + (int)threadIdx.x;
    if (gid >= a.xchunks) return;
    int ji = 0;
#pragma unroll
    for (int i = 1; i < FUSED_FRONT_MAXX; ++i)
      if (i < a.nxjob && gid >= a.xjob[i].chunk_begin) ji = i;
    shadow_chunk(a.xjob[ji], gid - a.xjob[ji].chunk_begin);
    return;
  }
  // split3 (the KG rows' launch in front of the one-launch RG forward): three blocks per tile, one in-projection pass each (the
  // projection is repeated) -- a third of the serial chain per block for rows that are few and on the critical path
  const int blk = a.split3 ? (int)blockIdx.x / 3 : (int)blockIdx.x;
  const int mypass = a.split3 ? (int)blockIdx.x - 3 * blk : 0;
  const int p_begin = a.split3 ? mypass : 0, p_end = a.split3 ? mypass + 1 : 3;
  const FrontStream& S = a.s[blk >= a.s[1].tile_begin ? 1 : 0];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = (blk - S.tile_begin) * (32 * RT);
  const int nrows = min(32 * RT, S.M - row0);
  char* bufR = smem + Cfg::BUFR; char* bufS = smem + Cfg::BUFS; char* bufX = bufS;
  float* cst = reinterpret_cast<float*>(smem + Cfg::CONSTS);
  stamp(a.stamps, 0);
  StageW<RT, 1, 8, 2, DEPTH> st0;
  st0.prefetch(reinterpret_cast<const u32x4*>(S.W0) + (size_t)((w8 >> 1) * (8 * 2) + (w8 & 1)) * 64 + lane);
  // input tile: fp32 -> bf16 (rows past the end cleared); the bf16 copy is also the weight-gradient operand X16
  constexpr int XIT = (RT * 256 + NTH - 1) / NTH;
  float4 xv[XIT][4];
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int idx = min(tid + NTH * it, RT * 256 - 1), r = idx >> 3, c = idx & 7;
    const float4* src = reinterpret_cast<const float4*>(S.X + (size_t)(row0 + min(r, nrows - 1)) * 128 + 16 * c);
#pragma unroll
    for (int q = 0; q < 4; ++q) xv[it][q] = src[q];
  }
  {
    const float* csrc = tid < 64 ? S.b0 + 4 * tid : (tid < 128 ? S.bq + 4 * (tid - 64) : S.bkv + 4 * (tid - 128));
    if (tid < 256) *reinterpret_cast<float4*>(cst + 4 * tid) = *reinterpret_cast<const float4*>(csrc);
  }
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int idx = tid + NTH * it, r = idx >> 3, c = idx & 7;
    if (idx < RT * 256) {
      const bool ok = r < nrows;
      const float4 v0 = xv[it][0], v1 = xv[it][1], v2 = xv[it][2], v3 = xv[it][3];
      u32x4 p0 = u32x4{pack2(v0.x, v0.y), pack2(v0.z, v0.w), pack2(v1.x, v1.y), pack2(v1.z, v1.w)};
      u32x4 p1 = u32x4{pack2(v2.x, v2.y), pack2(v2.z, v2.w), pack2(v3.x, v3.y), pack2(v3.z, v3.w)};
      if (!ok) { p0 = u32x4{0u, 0u, 0u, 0u}; p1 = p0; }
      *reinterpret_cast<u32x4*>(bufX + r * PX + 32 * c) = p0;
      *reinterpret_cast<u32x4*>(bufX + r * PX + 32 * c + 16) = p1;
      if (a.save && ok && mypass == 0) {
        u32x4* d = reinterpret_cast<u32x4*>(S.X16 + (size_t)(row0 + r) * 128 + 16 * c);
        d[0] = p0; d[1] = p1;
      }
    }
  }
  __syncthreads();
  stamp(a.stamps, 1);
  // projection 128 -> 256: wave w8 owns features 32 w8 .. + 31
  StageW<RT, 1, 16, 6, DEPTH> st1;
  const u32x4* w1base = reinterpret_cast<const u32x4*>(S.W1) + (size_t)((w8 >> 1) * (16 * 6) + 3 * (w8 & 1)) * 64 + lane;
  {
    f32x16 acc[RT][1];
    st0.template run<true, true>(bufX + l31 * PX + 16 * h, 32 * PX, acc);
    st1.prefetch(w1base + 64 * p_begin);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = 32 * w8 + 8 * g + 4 * h;
      const float4 bv = *reinterpret_cast<const float4*>(cst + c0);
#pragma unroll
      for (int s = 0; s < RT; ++s)
        *reinterpret_cast<u32x2*>(bufR + (32 * s + l31) * PR + 2 * c0) =
            u32x2{pack2(acc[s][0][4 * g] + bv.x, acc[s][0][4 * g + 1] + bv.y), pack2(acc[s][0][4 * g + 2] + bv.z, acc[s][0][4 * g + 3] + bv.w)};
    }
  }
  __syncthreads();                                             // R tile complete; every wave is done with the input tile
  stamp(a.stamps, 2);
  // in-projections 256 -> 768: pass p, feature tile tg = 3 w8 + p: tiles 0..7 = q (scaled), 8..23 = k' | v'
  char* strip = bufS + w8 * (32 * RT * PS);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    if (p < p_begin || p >= p_end) continue;                     // (block-uniform)
    f32x16 acc[RT][1];
    st1.template run<true, true>(bufR + l31 * PR + 16 * h, 32 * PR, acc);
    if (p + 1 < p_end) st1.prefetch(w1base + 64 * (p + 1));
    const int tg = 3 * w8 + p;                                   // wave-uniform
    const float sc = tg < 8 ? a.qscale : 1.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 8 * g + 4 * h;
      const float4 bv = *reinterpret_cast<const float4*>(cst + 256 + 32 * tg + c);
#pragma unroll
      for (int s = 0; s < RT; ++s)
        *reinterpret_cast<u32x2*>(strip + (32 * s + l31) * PS + 2 * c) =
            u32x2{pack2((acc[s][0][4 * g] + bv.x) * sc, (acc[s][0][4 * g + 1] + bv.y) * sc),
                  pack2((acc[s][0][4 * g + 2] + bv.z) * sc, (acc[s][0][4 * g + 3] + bv.w) * sc)};
    }
    us16* dbase = tg < 8 ? S.Q16 + (size_t)row0 * 256 + 32 * tg : S.KV16 + (size_t)row0 * 512 + 32 * (tg - 8);
    const int ld = tg < 8 ? 256 : 512;
#pragma unroll
    for (int it = 0; it < 2 * RT; ++it) {                       // the strip's rows: 4 lanes x 16 bytes each (the wave's own LDS writes: no barrier)
      const int idx = lane + 64 * it, r = idx >> 2, k = idx & 3;
      const u32x4 v = *reinterpret_cast<const u32x4*>(strip + r * PS + 16 * k);
      if (r < nrows && !(a.exp & 1)) *reinterpret_cast<u32x4*>(dbase + (size_t)r * ld + 8 * k) = v;
    }
    if (p == 0) stamp(a.stamps, 3);
    if (p == 1) stamp(a.stamps, 4);
  }
  stamp(a.stamps, 5);
  if (mypass == 0) copy_out<5, true>(bufR, PR, 0, S.R16, 256, row0, nrows, 32 * RT);
  for (int zi = 0; zi < a.nzero; ++zi) {                    // (behind the tile's own stores: nothing of this block waits for them)
    u32x4* z = static_cast<u32x4*>(a.zero_ptr[zi]);
    const unsigned n16 = a.zero_bytes[zi] >> 4;
    for (unsigned i = blockIdx.x * NTH + tid; i < n16; i += (unsigned)a.xblock0 * NTH) z[i] = u32x4{0u, 0u, 0u, 0u};      // (the tile blocks only)
  }
  stamp(a.stamps, 7);
}

// ------------------------------------------------------------------------------------------------ forward, back half
// Softmax over the <= 16 keys of one RG row: S holds the (pre-scaled) scores of keys acc_row(i, h), i < 8, in this lane
// and the other 8 keys in lane ^ 32.  (The same code as fused_rows.hip: backward recomputes these probabilities.)
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

constexpr int PART_FLOATS = 16 + 16 + 16 * 32;      // per (segment, head): max[16], sum[16], Z[16][32]
static_assert(PART_FLOATS == FUSED_PART_FLOATS, "fused_rows.h");

template <int RT> struct BackCfg {
  static constexpr int TILE = 32 * RT * PR;
  static constexpr int VS = RT * 16 * PV;                       // RG->KG: value rows of each sub-tile's sample
  static constexpr int VT = NW * 2048;                          // KG partial: per-wave V2 chunk [32][32] bf16, then its Z [16][32] fp32
  static constexpr int CMB = (FUSED_WIDE_MAXSEG + 1) * 512;     // last arriver: scale factors [nseg][128] + 1/L [128]
  static constexpr int SCR0 = VS + VT > TILE ? VS + VT : TILE;
  static constexpr int SCR = SCR0 > CMB ? SCR0 : CMB;           // bufY, aliased with the attention scratch and the combine's tables
  static constexpr int BUFO = 0, BUFY = TILE, RED = BUFY + SCR, RED_BYTES = 2 * NW * 32 * RT * 4, CST = RED + RED_BYTES, FLAG = CST + 1280 * 4, LDS = FLAG + 64;
};

struct Sub { int b; int row0; int nr; float inv_n; };          // one 32-row sub-tile (wave-uniform)

// a stream's epilogue constants -> LDS: bo [256] | ln_g [256] | ln_b [256] | b1 [512]  (a global load inside an epilogue stalls the wave 1-2 us)
__device__ __forceinline__ void load_consts(const BackStream& S, float* cst) {
  const int tid = threadIdx.x;
  if (tid < 320) {
    const float* src = tid < 64 ? S.bo + 4 * tid : (tid < 128 ? S.ln_g + 4 * (tid - 64) : (tid < 192 ? S.ln_b + 4 * (tid - 128) : S.b1 + 4 * (tid - 192)));
    *reinterpret_cast<float4*>(cst + 4 * tid) = *reinterpret_cast<const float4*>(src);
  }
}

// ---- out-projection + residual -> LayerNorm -> FFN layer 0 (+ReLU, dropout) -> pooled sums, for RTC sub-tiles whose attention
// output (bf16) sits in bufO.  Wave w8 owns features 32 w8.. of the 256-wide layers and 64 w8.. of the FFN layer.
// FUSEDIN (the one-launch RG forward): the attention output sits in the waves' strips [head][row][32 features] (pitch PS) at bufO,
// and the residual rows are the R tile in LDS at bufY, which the LayerNorm output then overwrites in place (each wave reads and
// writes only its own 32 columns); the residual R16 is saved from here.
template <int RTC, int DEPTH, bool DROP, bool SAVE, bool STAMPS, bool FUSEDIN = false>
__device__ __forceinline__ void chain_tiles(const BackArgs& a, const BackStream& S, const Sub (&sub)[RTC], char* bufO, char* bufY, float* red, const float* cst,
                                            int w8, int lane, StageW<RTC, 1, 16, 2, DEPTH>& sto) {
  const int l31 = lane & 31, h = lane >> 5;
  StageW<RTC, 2, 16, 4, DEPTH> stf;
  {
    // residual rows: issued in front of the out-projection so that they land while it runs
    u32x2 rv[RTC][4];
#pragma unroll
    for (int s = 0; s < RTC; ++s) {
      const size_t rrow = (size_t)sub[s].row0 + max(0, min(l31, sub[s].nr - 1));
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (!FUSEDIN) rv[s][g] = *reinterpret_cast<const u32x2*>(S.R16 + rrow * 256 + 32 * w8 + 8 * g + 4 * h);
    }
    f32x16 acc[RTC][1];
    if (FUSEDIN)
      sto.template run_f<true, true>([&](int s, int ks) { return *reinterpret_cast<const bf16x8*>(bufO + (ks >> 1) * (32 * RTC * PS) + (32 * s + l31) * PS + 32 * (ks & 1) + 16 * h); }, acc);
    else
      sto.template run<true, true>(bufO + l31 * PR + 16 * h, 32 * PR, acc);
    if (STAMPS) stamp(a.stamps, 5);
    if (FUSEDIN) {
#pragma unroll
      for (int s = 0; s < RTC; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          rv[s][g] = *reinterpret_cast<const u32x2*>(bufY + (32 * s + l31) * PR + 2 * (32 * w8 + 8 * g + 4 * h));
          if (SAVE && l31 < sub[s].nr) *reinterpret_cast<u32x2*>(const_cast<us16*>(S.R16) + ((size_t)sub[s].row0 + l31) * 256 + 32 * w8 + 8 * g + 4 * h) = rv[s][g];
        }
    }
    stf.prefetch(reinterpret_cast<const u32x4*>(S.W1) + (size_t)((w8 >> 1) * (16 * 4) + 2 * (w8 & 1)) * 64 + lane);   // (flows during the LayerNorm)
    float u[RTC][16];
    float part[RTC];
#pragma unroll
    for (int s = 0; s < RTC; ++s) {
      part[s] = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 32 * w8 + 8 * g + 4 * h;
        const float4 bv = *reinterpret_cast<const float4*>(cst + c0);
        float* o = u[s] + 4 * g;
        o[0] = acc[s][0][4 * g] + bv.x + bf_lo(rv[s][g].x); o[1] = acc[s][0][4 * g + 1] + bv.y + bf_hi(rv[s][g].x);
        o[2] = acc[s][0][4 * g + 2] + bv.z + bf_lo(rv[s][g].y); o[3] = acc[s][0][4 * g + 3] + bv.w + bf_hi(rv[s][g].y);
        part[s] += (o[0] + o[1]) + (o[2] + o[3]);
      }
    }
    // row totals: the lane's 16 values + the other lane half = this wave's 32 features; then the 8 waves through LDS
    auto row_total = [&](float (&p)[RTC], int slot) {
#pragma unroll
      for (int s = 0; s < RTC; ++s) {
        const float v = p[s] + __shfl_xor(p[s], 32, 64);
        if (h == 0) red[(slot * NW + w8) * (32 * RTC) + 32 * s + l31] = v;
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < RTC; ++s) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[(slot * NW + w) * (32 * RTC) + 32 * s + l31];
        p[s] = t;
      }
    };
    row_total(part, 0);
    if (STAMPS) stamp(a.stamps, 6);
    float sq[RTC];
#pragma unroll
    for (int s = 0; s < RTC; ++s) {
      const float mean = part[s] * (1.0f / 256.0f);
      sq[s] = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { u[s][i] -= mean; sq[s] = fmaf(u[s][i], u[s][i], sq[s]); }
    }
    row_total(sq, 1);
    if (STAMPS) stamp(a.stamps, 7);
    // (both barriers of row_total are behind every wave's out-projection MFMAs: bufY -- the attention scratch -- is free)
#pragma unroll
    for (int s = 0; s < RTC; ++s) {
      const float rstd = 1.0f / sqrtf(sq[s] * (1.0f / 256.0f) + 1e-5f);
      const bool rok = l31 < sub[s].nr;
      float ys[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 32 * w8 + 8 * g + 4 * h;
        const float4 gm = *reinterpret_cast<const float4*>(cst + 256 + c0), bt = *reinterpret_cast<const float4*>(cst + 512 + c0);
        float* o = u[s] + 4 * g;
        const float x0 = o[0] * rstd, x1 = o[1] * rstd, x2 = o[2] * rstd, x3 = o[3] * rstd;
        const float4 y = make_float4(x0 * gm.x + bt.x, x1 * gm.y + bt.y, x2 * gm.z + bt.z, x3 * gm.w + bt.w);
        *reinterpret_cast<u32x2*>(bufY + (32 * s + l31) * PR + 2 * c0) = u32x2{pack2(y.x, y.y), pack2(y.z, y.w)};
        ys[4 * g] = rok ? y.x : 0.f; ys[4 * g + 1] = rok ? y.y : 0.f; ys[4 * g + 2] = rok ? y.z : 0.f; ys[4 * g + 3] = rok ? y.w : 0.f;
        if (SAVE && rok)                                      // normalised LayerNorm input, straight from the registers (16 bytes per row and lane pair)
          *reinterpret_cast<u32x2*>(S.XH16 + ((size_t)sub[s].row0 + l31) * 256 + c0) = u32x2{pack2(x0, x1), pack2(x2, x3)};
      }
      if (SAVE && w8 == 0 && h == 0 && rok) S.rstd[(size_t)sub[s].row0 + l31] = rstd;
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
      if ((l31 & 1) == 0 && sub[s].nr > 0) {
        const int i16 = (l31 >> 1) & 15;
        atomicAdd(S.Ymean + (size_t)sub[s].b * 256 + 32 * w8 + 8 * (i16 >> 2) + 4 * h + (i16 & 3), tot * sub[s].inv_n);
      }
    }
  }
  if (STAMPS) stamp(a.stamps, 8);
  __syncthreads();
  if (STAMPS) stamp(a.stamps, 9);
  // ---- FFN layer 0 + ReLU + dropout, pooled over the rows (lane = feature; wave w8: features 64 w8 .. + 63)
  {
    f32x16 acc[RTC][2];
    stf.template run<false, true>(bufY + l31 * PR + 16 * h, 32 * PR, acc);
    if (STAMPS) stamp(a.stamps, 10);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int f = 64 * w8 + 32 * t + l31;
      const float bias = cst[768 + f];
#pragma unroll
      for (int s = 0; s < RTC; ++s) {
        const int nrows = sub[s].nr;
        if (nrows <= 0) continue;                             // (wave-uniform)
        float colsum = 0.f;
        uint32_t wlo = 0u, whi = 0u;
        if (nrows >= 32) {                                    // (wave-uniform) full sub-tiles -- the common case -- carry no row masks
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            float v = fmaxf(acc[s][t][i] + bias, 0.f);
            if (DROP) v *= drop_mult(a.drop, S.site_ffn, (uint32_t)(sub[s].row0 + acc_row(i, h)) * 512u + (uint32_t)f);
            colsum += v;
            if (SAVE) {
              const unsigned long long bal = __ballot(v > 0.f);
              if (lane == i) { wlo = (uint32_t)bal; whi = (uint32_t)(bal >> 32); }
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (8 * (i >> 2) >= nrows) continue;              // (block-uniform) a register group whose rows are all past the tile's end
            const int row = acc_row(i, h);
            float v = fmaxf(acc[s][t][i] + bias, 0.f);
            if (DROP) v *= drop_mult(a.drop, S.site_ffn, (uint32_t)(sub[s].row0 + row) * 512u + (uint32_t)f);
            if (row >= nrows) v = 0.f;
            colsum += v;
            if (SAVE) {
              const unsigned long long bal = __ballot(v > 0.f);
              if (lane == i) { wlo = (uint32_t)bal; whi = (uint32_t)(bal >> 32); }
            }
          }
        }
        colsum += __shfl_xor(colsum, 32, 64);
        if (h == 0) atomicAdd(S.Hmean + (size_t)sub[s].b * 512 + f, colsum * sub[s].inv_n);
        if (SAVE && lane < 16) {                              // lane i holds the words of rows acc_row(i, 0) and acc_row(i, 1), features 32 (2 w8 + t) ..
          const int ra = acc_row(lane, 0);
          if (ra < nrows) S.mask[((size_t)sub[s].row0 + ra) * 16 + 2 * w8 + t] = wlo;
          if (ra + 4 < nrows) S.mask[((size_t)sub[s].row0 + ra + 4) * 16 + 2 * w8 + t] = whi;
        }
      }
    }
  }
  if (STAMPS) stamp(a.stamps, 11);
  if (SAVE) {
    __syncthreads();                                          // (every wave is done reading bufY)
#pragma unroll
    for (int s = 0; s < RTC; ++s) {
      if (FUSEDIN) {                                          // strips: head w8's 32 features of row r at strip w8, row r: 4 x 16 bytes
        const int r = lane >> 2, k = lane & 3;
        if (r < sub[s].nr) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int rr = r + 16 * hh;
            if (rr < sub[s].nr)
              store16_wt(S.O16 + ((size_t)sub[s].row0 + rr) * 256 + 32 * w8 + 8 * k, *reinterpret_cast<const u32x4*>(bufO + w8 * (32 * RTC * PS) + (32 * s + rr) * PS + 16 * k));
          }
        }
      } else {
        copy_out<5, true>(bufO + 32 * s * PR, PR, 0, S.O16, 256, sub[s].row0, sub[s].nr, 32);
      }
      copy_out<5, true>(bufY + 32 * s * PR, PR, 0, S.Y16, 256, sub[s].row0, sub[s].nr, 32);
    }
  }
}

// The block that completed sample b: combine the nseg partials of its KG->RG attention into the attention output (bf16, rows
// j < Nk of bufO; the other rows zero).  Segment k's partial sits at tile slot k == 0 ? t0 : (t0 / RT + k) RT.  Every read is a
// long-latency miss (other CUs wrote a moment ago): loops run in batches of independent, unconditional loads.
template <int RT, bool SAVE>
__device__ __forceinline__ void kg_combine(const BackArgs& a, char* scratch, char* bufO, int b, int t0, int nseg) {
  const int tid = threadIdx.x, Nk = a.Nk;
  float* sc = reinterpret_cast<float*>(scratch);           // [nseg][128] scale factors exp(m_s - M), then [128] 1 / L
  float* invL = sc + nseg * 128;
  auto seg_part = [&](int k) { return a.part + (size_t)(k == 0 ? t0 : (t0 / RT + k) * RT) * 8 * PART_FLOATS; };
  // thread (head = tid >> 6, query = (tid >> 2) & 15, features 8 (tid & 3) .. + 7): two 16-byte loads per segment
  const int hd = tid >> 6, j = (tid >> 2) & 15, f8 = tid & 3;
  f32x4 zv[4][2];
  auto load_z = [&](int s0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float* z = seg_part(min(s0 + k, nseg - 1)) + (size_t)hd * PART_FLOATS + 32 + j * 32 + 8 * f8;
      zv[k][0] = *reinterpret_cast<const f32x4*>(z); zv[k][1] = *reinterpret_cast<const f32x4*>(z + 4);
    }
  };
  load_z(0);
  if (tid < 128) {
    const int hd2 = tid >> 4, j2 = tid & 15;
    float M = -INFINITY, L = 0.f;
    for (int s0 = 0; s0 < nseg; s0 += 8) {
      float mv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) mv[k] = seg_part(min(s0 + k, nseg - 1))[(size_t)hd2 * PART_FLOATS + j2];
#pragma unroll
      for (int k = 0; k < 8; ++k) M = fmaxf(M, mv[k]);
    }
    for (int s0 = 0; s0 < nseg; s0 += 8) {
      float mv[8], lv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float* p = seg_part(min(s0 + k, nseg - 1)) + (size_t)hd2 * PART_FLOATS + j2; mv[k] = p[0]; lv[k] = p[16]; }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (s0 + k < nseg) { const float e = __expf(mv[k] - M); sc[(s0 + k) * 128 + tid] = e; L = fmaf(lv[k], e, L); }
    }
    invL[tid] = 1.0f / L;
    if (SAVE && a.lse2 && j2 < Nk) { float* o = a.lse2 + (((size_t)b * 8 + hd2) * 16 + j2) * 2; o[0] = M; o[1] = L; }
  }
  __syncthreads();
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  for (int s0 = 0; s0 < nseg; s0 += 4) {
    if (s0 > 0) load_z(s0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (s0 + k < nseg) { const float e = sc[(s0 + k) * 128 + hd * 16 + j]; acc0 += zv[k][0] * e; acc1 += zv[k][1] * e; }
  }
  if (j < Nk) {
    const float il = invL[hd * 16 + j];
    *reinterpret_cast<u32x4*>(bufO + j * PR + 2 * (32 * hd + 8 * f8)) =
        u32x4{pack2(acc0[0] * il, acc0[1] * il), pack2(acc0[2] * il, acc0[3] * il), pack2(acc1[0] * il, acc1[1] * il), pack2(acc1[2] * il, acc1[3] * il)};
  }
}

template <int RT, int DEPTH, bool DROP, bool SAVE>
__global__ __launch_bounds__(NTH, 2) void back8_kernel(const BackArgs a) {
  using Cfg = BackCfg<RT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, Nk = a.Nk;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);        // = the attention head this wave owns
  char* bufO = smem + Cfg::BUFO; char* bufY = smem + Cfg::BUFY;
  char* Vs = bufY; char* Vt = bufY + Cfg::VS + w8 * 2048;
  float* red = reinterpret_cast<float*>(smem + Cfg::RED);
  float* cst = reinterpret_cast<float*>(smem + Cfg::CST);
  int* flag = reinterpret_cast<int*>(smem + Cfg::FLAG);
  const int g0 = (int)blockIdx.x * RT;
  Sub sub[RT];
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    int4 td = make_int4(-1, 0, 0, 0);
    if (g0 + s < a.rg_tiles_max) td = a.tile_desc[g0 + s];
    // (wave-uniform by construction; readfirstlane makes it provable: scalar address arithmetic and branches from here on)
    const int tb = __builtin_amdgcn_readfirstlane(td.x);
    sub[s].b = tb < 0 ? 0 : tb; sub[s].row0 = tb < 0 ? 0 : __builtin_amdgcn_readfirstlane(td.y); sub[s].nr = tb < 0 ? 0 : __builtin_amdgcn_readfirstlane(td.z);
    sub[s].inv_n = tb < 0 ? 0.f : __int_as_float(__builtin_amdgcn_readfirstlane(td.w));
  }
  if (sub[0].nr == 0) return;                                     // (tiles are dense from 0: the whole group is past the end)
  stamp(a.stamps, 0);
  const BackStream& S = a.s[0];
  StageW<RT, 1, 16, 2, DEPTH> sto;
  sto.prefetch(reinterpret_cast<const u32x4*>(S.Wo) + (size_t)((w8 >> 1) * (16 * 2) + (w8 & 1)) * 64 + lane);      // (flows during the attention)
  // ---- every load of both attention directions (head w8) up front
  bf16x8 k2f[RT][2], q2f[RT][2], qf[RT][2], kf[RT][2];
  u32x4 vv[RT][2];
  int nseg_of[RT];                                                // segments (blocks) of sub-tile s's sample: for the arrival tickets
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    const size_t rrow = (size_t)sub[s].row0 + max(0, min(l31, sub[s].nr - 1));
    const us16* k2p = a.KV2_16 + rrow * 512 + 32 * w8 + 8 * h;                 // RG keys of the tile's own rows (A operand: lane = key row)
    const us16* q2p = a.Q2_16 + ((size_t)sub[s].b * Nk + min(l31, Nk - 1)) * 256 + 32 * w8 + 8 * h;   // KG queries (B operand: lane = query)
    const us16* qp = a.Q16 + rrow * 256 + 32 * w8 + 8 * h;                     // RG queries (B operand: lane = row)
    const int jk = l31 & 15;
    const us16* kp = a.KV16 + ((size_t)sub[s].b * Nk + min(jk, Nk - 1)) * 512 + 32 * w8 + 8 * h;   // KG keys (A operand: lane = key)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      k2f[s][k] = as_frag(*reinterpret_cast<const u32x4*>(k2p + 16 * k));
      q2f[s][k] = as_frag(*reinterpret_cast<const u32x4*>(q2p + 16 * k));
      qf[s][k] = as_frag(*reinterpret_cast<const u32x4*>(qp + 16 * k));
      u32x4 kv = *reinterpret_cast<const u32x4*>(kp + 16 * k);
      if (jk >= Nk) kv = u32x4{0u, 0u, 0u, 0u};
      kf[s][k] = as_frag(kv);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {                             // this head's value chunk of the tile's rows: slot id -> row id >> 2, 16-byte chunk id & 3
      const int id = lane + 64 * i;
      vv[s][i] = *reinterpret_cast<const u32x4*>(a.KV2_16 + ((size_t)sub[s].row0 + max(0, min(id >> 2, sub[s].nr - 1))) * 512 + 256 + 32 * w8 + 8 * (id & 3));
    }
    const int t0 = __builtin_amdgcn_readfirstlane(a.tile_off[sub[s].b]), t1 = __builtin_amdgcn_readfirstlane(a.tile_off[sub[s].b + 1]);
    nseg_of[s] = (t1 - 1) / RT - t0 / RT + 1;
  }
  load_consts(S, cst);
#pragma unroll
  for (int s = 0; s < RT; ++s) {   // the sample's KG value rows [Nk][256] -> image (rows Nk..15 cleared): one 16-byte chunk per thread
    const int j = tid >> 5, ch = tid & 31;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (j < Nk) v = *reinterpret_cast<const u32x4*>(a.KV16 + ((size_t)sub[s].b * Nk + j) * 512 + 256 + 8 * ch);
    *reinterpret_cast<u32x4*>(Vs + s * (16 * PV) + j * PV + 16 * ch) = v;
  }
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  // ---- KG->RG attention, this block's rows as keys: one flash-style partial per run of sub-tiles of the same sample
  {
    f32x16 S2[RT];
    float mx[RT];
#pragma unroll
    for (int s = 0; s < RT; ++s) {
      S2[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k2f[s][0], q2f[s][0], zero16(), 0, 0, 0);   // [key row of the sub-tile][query]: lane = query
      S2[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k2f[s][1], q2f[s][1], S2[s], 0, 0, 0);
      float m = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) m = acc_row(i, h) < sub[s].nr ? fmaxf(m, S2[s][i]) : m;
      mx[s] = fmaxf(m, __shfl_xor(m, 32, 64));
    }
    float mseg[RT];
#pragma unroll
    for (int s = 0; s < RT; ++s) {
      mseg[s] = mx[s];
#pragma unroll
      for (int s2 = 0; s2 < RT; ++s2)
        if (s2 != s && sub[s2].b == sub[s].b && sub[s2].nr > 0) mseg[s] = fmaxf(mseg[s], mx[s2]);     // (wave-uniform conditions)
    }
    f32x16 Z = zero16();
    float L = 0.f;
    int seg_first = g0;
#pragma unroll
    for (int s = 0; s < RT; ++s) {
      if (sub[s].nr <= 0) continue;                           // (wave-uniform)
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(Vt + 16 * (lane + 64 * i)) = vv[s][i];      // [32 rows][64 bytes], linear
      float e[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = acc_row(i, h);
        e[i] = row < sub[s].nr ? __expf(S2[s][i] - mseg[s]) : 0.f;
        L += e[i];
        if (DROP) e[i] *= drop_mult(a.drop, SITE_ATTN_KG2RG, ((uint32_t)(sub[s].row0 + row) * 8u + (uint32_t)w8) * (uint32_t)Nk + (uint32_t)l31);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const bf16x8 ef = as_frag(u32x4{pack2(e[8 * k], e[8 * k + 1]), pack2(e[8 * k + 2], e[8 * k + 3]),
                                        pack2(e[8 * k + 4], e[8 * k + 5]), pack2(e[8 * k + 6], e[8 * k + 7])});
        const char* vp = Vt + (16 * k + 4 * h + q4) * 64 + 2 * (16 * g1 + 4 * p4);
        const bf16x8 vf = join(lds_tr16(vp), lds_tr16(vp + 8 * 64));
        Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ef, vf, Z, 0, 0, 0);      // Z[query][feature] += E^T . V
      }
      const bool seg_end = s == RT - 1 || sub[s + 1].nr <= 0 || sub[s + 1].b != sub[s].b;      // (wave-uniform)
      if (seg_end) {
        L += __shfl_xor(L, 32, 64);
        float* part = a.part + ((size_t)seg_first * 8 + w8) * PART_FLOATS;
        if (lane < 16) {
          __hip_atomic_store(part + lane, mseg[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(part + 16 + lane, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // Z [16 queries][32 features] through this wave's value-chunk scratch so that it leaves as 16-byte write-through stores
        float* zt = reinterpret_cast<float*>(Vt);
#pragma unroll
        for (int i = 0; i < 8; ++i) zt[acc_row(i, h) * 32 + l31] = Z[i];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int idx = lane + 64 * r;
          const f32x4 v = *reinterpret_cast<const f32x4*>(zt + 4 * idx);
          float* dst = part + 32 + 4 * idx;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
        }
        Z = zero16(); L = 0.f; seg_first = g0 + s + 1;
      }
    }
  }
  stamp(a.stamps, 1);
  __syncthreads();                                                // KG value images and the constants are in LDS
  stamp(a.stamps, 2);
  // ---- RG->KG attention: rows of sub-tile s against the Nk keys of its sample; output (bf16) -> bufO
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    f32x16 Sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s][0], qf[s][0], zero16(), 0, 0, 0);     // S^T[key][row]: lane = row
    Sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s][1], qf[s][1], Sc, 0, 0, 0);
    float e[8];
    rg_softmax(Sc, h, Nk, e);
    const uint32_t ibase = ((uint32_t)(sub[s].row0 + l31) * 8u + (uint32_t)w8) * (uint32_t)Nk;
    if (DROP) {
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] *= drop_mult(a.drop, SITE_ATTN_RG2KG, ibase + (uint32_t)acc_row(i, h));
    }
    const bf16x8 pf = as_frag(u32x4{pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])});
    const char* vp = Vs + s * (16 * PV) + (4 * h + q4) * PV + 2 * (32 * w8 + 16 * g1 + 4 * p4);
    const bf16x8 vf = join(lds_tr16(vp), lds_tr16(vp + 8 * PV));
    const f32x16 O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, zero16(), 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<u32x2*>(bufO + (32 * s + l31) * PR + 2 * (32 * w8 + 8 * g + 4 * h)) =
          u32x2{pack2(O[4 * g], O[4 * g + 1]), pack2(O[4 * g + 2], O[4 * g + 3])};
  }
  // ---- arrival: one ticket per (sample, segment) this block wrote a partial for (hand-off: write-through partial stores, every
  // storing wave's drain, the block's barrier, a relaxed agent-scope ticket).  Drawn here, read at the end of the block: whoever
  // drew a sample's last ticket combines its partials and runs the KG rows' chain behind its own.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(a.stamps, 3);
  __syncthreads();                                                // bufO complete; every wave's partial stores acknowledged
  int tk[RT];
  if (tid == 0) {
    int prev = -1;
#pragma unroll
    for (int s = 0; s < RT; ++s) {
      tk[s] = -1;
      if (sub[s].nr > 0 && sub[s].b != prev) {
        prev = sub[s].b;
        tk[s] = __hip_atomic_fetch_add(a.tickets + prev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  stamp(a.stamps, 4);
  chain_tiles<RT, DEPTH, DROP, SAVE, true>(a, S, sub, bufO, bufY, red, cst, w8, lane, sto);
  stamp(a.stamps, 12);
  if (tid == 0) {
#pragma unroll
    for (int s = 0; s < RT; ++s) flag[s] = (tk[s] >= 0 && tk[s] == nseg_of[s] - 1) ? nseg_of[s] : -1;
  }
  __syncthreads();
  int todo[RT];
#pragma unroll
  for (int s = 0; s < RT; ++s) todo[s] = flag[s];
  bool any = false;
#pragma unroll
  for (int s = 0; s < RT; ++s) any = any || todo[s] > 0;
  if (!any) return;                                               // (block-uniform)
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const BackStream& K = a.s[1];
#pragma unroll 1
  for (int s = 0; s < RT; ++s) {
    const int nseg = __builtin_amdgcn_readfirstlane(todo[s]);
    if (nseg <= 0) continue;
    const int b = sub[s].b;
    __syncthreads();                                              // (the acquire above / the previous sample's chain is done with the buffers)
    for (int c = tid; c < 32 * PR / 16; c += NTH) reinterpret_cast<u32x4*>(bufO)[c] = u32x4{0u, 0u, 0u, 0u};   // rows >= Nk stay zero
    StageW<1, 1, 16, 2, DEPTH> stk;
    stk.prefetch(reinterpret_cast<const u32x4*>(K.Wo) + (size_t)((w8 >> 1) * (16 * 2) + (w8 & 1)) * 64 + lane);
    load_consts(K, cst);
    __syncthreads();
    kg_combine<RT, SAVE>(a, bufY, bufO, b, a.tile_off[b], nseg);
    __syncthreads();
    stamp(a.stamps, 13);
    const Sub ksub[1] = {Sub{b, b * Nk, Nk, 1.0f / (float)Nk}};
    chain_tiles<1, DEPTH, DROP, SAVE, false>(a, K, ksub, bufO, bufY, red, cst, w8, lane, stk);
    stamp(a.stamps, 14);
  }
}

// ------------------------------------------------------------------------------------------------ forward, RG rows in ONE launch
// x -> R -> [q | k2 | v2] -> both attention directions -> out-projection + residual -> LayerNorm -> FFN layer 0 -> pooled sums for
// 32 RT rows, without a global round trip in between: what the front / back pair hands over through HBM (Q16, KV2_16, R16: 2 KB
// per row written, then read back behind a kernel boundary) stays in registers and LDS here.  The in-projection passes give
// wave w8 the q, k2 and v2 features of ITS head (feature tile 8 p + w8 of the [768 x 256] shadow), so that
//   * q^T (lane = row, registers = the head's 32 features) is, packed to bf16, the B operand of S^T = K_h Q_h^T as it stands;
//   * k2^T is the A operand of the KG->RG scores S2 = K2 Q2^T as it stands (both with the k order of an accumulator tile:
//     element j of lane half h = feature 16 kk + 8 (j >> 2) + 4 h + (j & 3); the other operand is loaded in that order);
//   * the v2 pass runs with the operands swapped (lane = feature, registers = rows), which makes it the B operand of Z += E^T V2.
// Only the RG->KG attention output goes through LDS (the wave's strip: the out-projection reads all heads), and the 16 KG value
// rows of a sample pass through a 1 KB scratch for the transposing read.  LDS: R / Y tile (in place) 66 KB, strips 80 KB.
// Needs the KG rows' projections (Q2_16, KV16) from an earlier launch (the front kernel on the KG stream alone).
struct RgFwdArgs { FrontStream f; BackArgs b; float qscale; };

template <int RT> struct RgCfg {
  static constexpr int BUFR = 0, STR = 32 * RT * PR, STRIP = 32 * RT * PS;
  static constexpr int STR_BYTES0 = NW * STRIP, XT = 32 * RT * PX, KGS = 32 * PR + (FUSED_WIDE_MAXSEG + 1) * 512;     // strips | input tile | KG chain: one attention tile + the combine's tables
  static constexpr int STR_BYTES1 = STR_BYTES0 > XT ? STR_BYTES0 : XT, STR_BYTES = STR_BYTES1 > KGS ? STR_BYTES1 : KGS;
  static constexpr int RED = STR + STR_BYTES, RED_BYTES = 2 * NW * 32 * RT * 4, CST = RED + RED_BYTES, FLAG = CST + 1280 * 4, LDS = FLAG + 64;
};

template <int RT, bool DROP, bool SAVE>
__global__ __launch_bounds__(NTH, 2) void rgfwd_kernel(const RgFwdArgs g) {
  using Cfg = RgCfg<RT>;
  constexpr int DEPTH = 8, DIN = 12;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FrontStream& F = g.f;
  const BackArgs& a = g.b;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, Nk = a.Nk;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);        // = the attention head this wave owns
  char* bufR = smem + Cfg::BUFR; char* strips = smem + Cfg::STR; char* bufX = strips;
  char* strip = strips + w8 * Cfg::STRIP;
  float* red = reinterpret_cast<float*>(smem + Cfg::RED);
  float* cst = reinterpret_cast<float*>(smem + Cfg::CST);
  int* flag = reinterpret_cast<int*>(smem + Cfg::FLAG);
  const int g0 = (int)blockIdx.x * RT;
  Sub sub[RT];
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    int4 td = make_int4(-1, 0, 0, 0);
    if (g0 + s < a.rg_tiles_max) td = a.tile_desc[g0 + s];
    const int tb = __builtin_amdgcn_readfirstlane(td.x);
    sub[s].b = tb < 0 ? 0 : tb; sub[s].row0 = tb < 0 ? 0 : __builtin_amdgcn_readfirstlane(td.y); sub[s].nr = tb < 0 ? 0 : __builtin_amdgcn_readfirstlane(td.z);
    sub[s].inv_n = tb < 0 ? 0.f : __int_as_float(__builtin_amdgcn_readfirstlane(td.w));
  }
  if (sub[0].nr == 0) return;                                     // (tiles are dense from 0: the whole group is past the end)
  stamp(a.stamps, 0);
  StageW<RT, 1, 8, 2, DEPTH> st0;
  st0.prefetch(reinterpret_cast<const u32x4*>(F.W0) + (size_t)((w8 >> 1) * (8 * 2) + (w8 & 1)) * 64 + lane);
  // First touch of the KG-side rows this block will want two passes from now (the sample's keys, values and queries of head w8:
  // a TLB walk + an HBM read each, 3-5 us cold) and of its tile range: ONE dword per lane, issued beside the input rows so that
  // the latencies overlap; the real loads later hit L2.  (Holding the fragments in registers from here on instead made hipcc
  // spill them one by one behind a full wait each: 8 us.)
  uint32_t touch[RT];
  int toff[RT][2];
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    const int j = min(lane & 15, Nk - 1), what = lane >> 4;        // 0: key slice, 1: value slice, 2: query row, 3: key slice again
    const us16* tp = what == 2 ? a.Q2_16 + ((size_t)sub[s].b * Nk + j) * 256 + 32 * w8 : a.KV16 + ((size_t)sub[s].b * Nk + j) * 512 + (what == 1 ? 256 : 0) + 32 * w8;
    touch[s] = *reinterpret_cast<const uint32_t*>(tp);
    toff[s][0] = a.tile_off[sub[s].b]; toff[s][1] = a.tile_off[sub[s].b + 1];
  }
  // ---- input rows of the RT sub-tiles: fp32 -> bf16 tile (rows past a sub-tile's end cleared); the bf16 copy is the operand X16
  {
    constexpr int XIT = (RT * 256 + NTH - 1) / NTH;
    float4 xv[XIT][4];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int idx = min(tid + NTH * it, RT * 256 - 1), r = idx >> 3, c = idx & 7, sx = r >> 5;
      int srow = 0, snr = 0;
#pragma unroll
      for (int s = 0; s < RT; ++s) if (s == sx) { srow = sub[s].row0; snr = sub[s].nr; }
      const float4* src = reinterpret_cast<const float4*>(F.X + ((size_t)srow + max(0, min(r & 31, snr - 1))) * 128 + 16 * c);
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[it][q] = src[q];
    }
    __builtin_amdgcn_sched_barrier(0);                          // (issue order pinned: input rows, constants, then the late-use loads)
    // epilogue constants of the first two layers: b0 [256] | bq [256] | bkv [512]; the chain's constants replace them later
    float4 c0v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 256) c0v = *reinterpret_cast<const float4*>(tid < 64 ? F.b0 + 4 * tid : (tid < 128 ? F.bq + 4 * (tid - 64) : F.bkv + 4 * (tid - 128)));
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int idx = tid + NTH * it, r = idx >> 3, c = idx & 7, sx = r >> 5;
      if (idx < RT * 256) {
        int srow = 0, snr = 0;
#pragma unroll
        for (int s = 0; s < RT; ++s) if (s == sx) { srow = sub[s].row0; snr = sub[s].nr; }
        const bool ok = (r & 31) < snr;
        const float4 v0 = xv[it][0], v1 = xv[it][1], v2 = xv[it][2], v3 = xv[it][3];
        u32x4 p0 = u32x4{pack2(v0.x, v0.y), pack2(v0.z, v0.w), pack2(v1.x, v1.y), pack2(v1.z, v1.w)};
        u32x4 p1 = u32x4{pack2(v2.x, v2.y), pack2(v2.z, v2.w), pack2(v3.x, v3.y), pack2(v3.z, v3.w)};
        if (!ok) { p0 = u32x4{0u, 0u, 0u, 0u}; p1 = p0; }
        *reinterpret_cast<u32x4*>(bufX + r * PX + 32 * c) = p0;
        *reinterpret_cast<u32x4*>(bufX + r * PX + 32 * c + 16) = p1;
        if (SAVE && ok) {
          u32x4* d = reinterpret_cast<u32x4*>(F.X16 + ((size_t)srow + (r & 31)) * 128 + 16 * c);
          d[0] = p0; d[1] = p1;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < RT; ++s) asm volatile("" ::"v"(touch[s]));
    stamp(a.stamps, 9);
    if (tid < 256) *reinterpret_cast<float4*>(cst + 4 * tid) = c0v;
    stamp(a.stamps, 10);
  }
  __syncthreads();
  stamp(a.stamps, 1);
  // ---- projection 128 -> 256: wave w8 owns features 32 w8 .. + 31 of the R tile
  StageW<RT, 1, 16, 6, DIN> st1;
  auto w1tile = [&](int tg) { return reinterpret_cast<const u32x4*>(F.W1) + (size_t)((tg / 6) * (16 * 6) + tg % 6) * 64 + lane; };
  {
    f32x16 acc[RT][1];
    st0.template run<true, true>(bufX + l31 * PX + 16 * h, 32 * PX, acc);
    st1.prefetch(w1tile(8 + w8));
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int c0 = 32 * w8 + 8 * gq + 4 * h;
      const float4 bv = *reinterpret_cast<const float4*>(cst + c0);
#pragma unroll
      for (int s = 0; s < RT; ++s)
        *reinterpret_cast<u32x2*>(bufR + (32 * s + l31) * PR + 2 * c0) =
            u32x2{pack2(acc[s][0][4 * gq] + bv.x, acc[s][0][4 * gq + 1] + bv.y), pack2(acc[s][0][4 * gq + 2] + bv.z, acc[s][0][4 * gq + 3] + bv.w)};
    }
  }
  // the chain's epilogue constants: loaded now, parked in LDS once the in-projection biases have been read (after pass 2)
  const BackStream& S = a.s[0];
  float4 c1v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 320) c1v = *reinterpret_cast<const float4*>(tid < 64 ? S.bo + 4 * tid : (tid < 128 ? S.ln_g + 4 * (tid - 64) : (tid < 192 ? S.ln_b + 4 * (tid - 128) : S.b1 + 4 * (tid - 192))));
  __syncthreads();                                                // R tile complete; every wave is done with the input tile
  stamp(a.stamps, 2);
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  // ---- passes 0, 1: k2^T (lane = key row) and v2 (lane = feature) of head w8 -> the KG->RG partial of this block's rows
  // (first: its Z leaves through the wave's strip, which the RG->KG attention output occupies from pass 2 on).
  // The scores, their segment maxima and the exponentials are finished between the two passes (k2 and the queries die there);
  // what crosses the v2 pass are the bf16 exponentials (the A operand of Z += E^T V2) and the row sums.
  {
    u32x4 ef[RT][2];
    float Lsub[RT], mseg[RT];
    {
      // the sample's KG queries in accumulator k order (B operand: lane = query)
      u32x4 q2f[RT][2];
#pragma unroll
      for (int s = 0; s < RT; ++s) {
        const us16* q2p = a.Q2_16 + ((size_t)sub[s].b * Nk + min(l31, Nk - 1)) * 256 + 32 * w8 + 4 * h;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const u32x2 lo = *reinterpret_cast<const u32x2*>(q2p + 16 * kk), hi = *reinterpret_cast<const u32x2*>(q2p + 16 * kk + 8);
          q2f[s][kk] = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
      }
      f32x16 acc[RT][1];
      st1.template run<true, true>(bufR + l31 * PR + 16 * h, 32 * PR, acc);
      st1.prefetch(w1tile(16 + w8));
      stamp(a.stamps, 3);
      f32x16 S2[RT];
      float mx[RT];
#pragma unroll
      for (int s = 0; s < RT; ++s) {
        u32x4 k2f[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = acc[s][0][8 * kk + j] + cst[512 + 32 * w8 + acc_row(8 * kk + j, h)];
          k2f[kk] = u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
          if (SAVE && l31 < sub[s].nr) {
            us16* kd = F.KV16 + ((size_t)sub[s].row0 + l31) * 512 + 32 * w8 + 16 * kk + 4 * h;
            *reinterpret_cast<u32x2*>(kd) = u32x2{k2f[kk].x, k2f[kk].y};
            *reinterpret_cast<u32x2*>(kd + 8) = u32x2{k2f[kk].z, k2f[kk].w};
          }
        }
        S2[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(k2f[0]), as_frag(q2f[s][0]), zero16(), 0, 0, 0);   // [key row of the sub-tile][query]: lane = query
        S2[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(k2f[1]), as_frag(q2f[s][1]), S2[s], 0, 0, 0);
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) m = acc_row(i, h) < sub[s].nr ? fmaxf(m, S2[s][i]) : m;
        mx[s] = fmaxf(m, __shfl_xor(m, 32, 64));
      }
#pragma unroll
      for (int s = 0; s < RT; ++s) {
        mseg[s] = mx[s];
#pragma unroll
        for (int s2 = 0; s2 < RT; ++s2)
          if (s2 != s && sub[s2].b == sub[s].b && sub[s2].nr > 0) mseg[s] = fmaxf(mseg[s], mx[s2]);     // (wave-uniform conditions)
      }
#pragma unroll
      for (int s = 0; s < RT; ++s) {
        float e[16];
        Lsub[s] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = acc_row(i, h);
          e[i] = row < sub[s].nr ? __expf(S2[s][i] - mseg[s]) : 0.f;
          Lsub[s] += e[i];
          if (DROP) e[i] *= drop_mult(a.drop, SITE_ATTN_KG2RG, ((uint32_t)(sub[s].row0 + row) * 8u + (uint32_t)w8) * (uint32_t)Nk + (uint32_t)l31);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
          ef[s][k] = u32x4{pack2(e[8 * k], e[8 * k + 1]), pack2(e[8 * k + 2], e[8 * k + 3]), pack2(e[8 * k + 4], e[8 * k + 5]), pack2(e[8 * k + 6], e[8 * k + 7])};
      }
    }
    f32x16 vacc[RT][1];
    st1.template run<false, true>(bufR + l31 * PR + 16 * h, 32 * PR, vacc);      // same fragments, operands swapped
    st1.prefetch(w1tile(w8));
    stamp(a.stamps, 4);
    const float vbias = cst[768 + 32 * w8 + l31];
    f32x16 Z = zero16();
    float L = 0.f;
    int seg_first = g0;
    float* zt = reinterpret_cast<float*>(strip);                         // [16 queries][32 features] fp32 (the strip is free until pass 2)
#pragma unroll
    for (int s = 0; s < RT; ++s) {
      if (sub[s].nr <= 0) continue;                           // (wave-uniform)
      if (SAVE) {                                             // v2 rows (bf16): lane = feature, register i = row acc_row(i, h)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (acc_row(i, h) < sub[s].nr) F.KV16[((size_t)sub[s].row0 + acc_row(i, h)) * 512 + 256 + 32 * w8 + l31] = f2bf(vacc[s][0][i] + vbias);
      }
      L += Lsub[s];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const bf16x8 vf = as_frag(u32x4{pack2(vacc[s][0][8 * k] + vbias, vacc[s][0][8 * k + 1] + vbias), pack2(vacc[s][0][8 * k + 2] + vbias, vacc[s][0][8 * k + 3] + vbias),
                                        pack2(vacc[s][0][8 * k + 4] + vbias, vacc[s][0][8 * k + 5] + vbias), pack2(vacc[s][0][8 * k + 6] + vbias, vacc[s][0][8 * k + 7] + vbias)});
        Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ef[s][k]), vf, Z, 0, 0, 0);      // Z[query][feature] += E^T . V2
      }
      const bool seg_end = s == RT - 1 || sub[s + 1].nr <= 0 || sub[s + 1].b != sub[s].b;      // (wave-uniform)
      if (seg_end) {
        L += __shfl_xor(L, 32, 64);
        float* part = a.part + ((size_t)seg_first * 8 + w8) * PART_FLOATS;
        if (lane < 16) {
          __hip_atomic_store(part + lane, mseg[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(part + 16 + lane, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) zt[acc_row(i, h) * 32 + l31] = Z[i];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int idx = lane + 64 * r;
          const f32x4 v = *reinterpret_cast<const f32x4*>(zt + 4 * idx);
          float* dst = part + 32 + 4 * idx;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
        }
        Z = zero16(); L = 0.f; seg_first = g0 + s + 1;
      }
    }
  }
  // ---- pass 2: q^T of head w8 (lane = row) -> RG->KG attention straight from the accumulators -> O into the wave's strip
  stamp(a.stamps, 5);
  StageW<RT, 1, 16, 2, DEPTH> sto;
  {
    // the sample's KG keys in accumulator k order (A operand: lane = key) and its value rows [16][32 features of the head] (L2 hits by now)
    u32x4 kf[RT][2], vkg[RT];
#pragma unroll
    for (int s = 0; s < RT; ++s) {
      const int jk = l31 & 15;
      const us16* kp = a.KV16 + ((size_t)sub[s].b * Nk + min(jk, Nk - 1)) * 512 + 32 * w8 + 4 * h;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const u32x2 lo = *reinterpret_cast<const u32x2*>(kp + 16 * kk), hi = *reinterpret_cast<const u32x2*>(kp + 16 * kk + 8);
        kf[s][kk] = jk < Nk ? u32x4{lo.x, lo.y, hi.x, hi.y} : u32x4{0u, 0u, 0u, 0u};
      }
      const int j = lane >> 2, c = lane & 3;
      vkg[s] = u32x4{0u, 0u, 0u, 0u};
      if (j < Nk) vkg[s] = *reinterpret_cast<const u32x4*>(a.KV16 + ((size_t)sub[s].b * Nk + j) * 512 + 256 + 32 * w8 + 8 * c);
    }
    f32x16 acc[RT][1];
    st1.template run<true, true>(bufR + l31 * PR + 16 * h, 32 * PR, acc);
    // arrival, first half (see back8_kernel): this wave's partial stores are drained HERE -- they are a whole pass old, and nothing
    // younger is in flight yet -- rather than in front of the barrier below, where the wait would also sit out the next stream's prefetch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sto.prefetch(reinterpret_cast<const u32x4*>(S.Wo) + (size_t)((w8 >> 1) * (16 * 2) + (w8 & 1)) * 64 + lane);
    stamp(a.stamps, 6);
#pragma unroll
    for (int s = 0; s < RT; ++s) {
      char* scr = strip + 32 * s * PS;                          // 1 KB scratch for the transposing read: the sub-tile's own (not yet written) output rows
      u32x4 qf[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (acc[s][0][8 * kk + j] + cst[256 + 32 * w8 + acc_row(8 * kk + j, h)]) * g.qscale;
        qf[kk] = u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        if (SAVE && l31 < sub[s].nr) {                          // Q16: features 16 kk + 4 h .. + 3 and 16 kk + 8 + 4 h .. + 3 of the head
          us16* qd = F.Q16 + ((size_t)sub[s].row0 + l31) * 256 + 32 * w8 + 16 * kk + 4 * h;
          *reinterpret_cast<u32x2*>(qd) = u32x2{qf[kk].x, qf[kk].y};
          *reinterpret_cast<u32x2*>(qd + 8) = u32x2{qf[kk].z, qf[kk].w};
        }
      }
      f32x16 Sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(kf[s][0]), as_frag(qf[0]), zero16(), 0, 0, 0);     // S^T[key][row]: lane = row
      Sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(kf[s][1]), as_frag(qf[1]), Sc, 0, 0, 0);
      float e[8];
      rg_softmax(Sc, h, Nk, e);
      const uint32_t ibase = ((uint32_t)(sub[s].row0 + l31) * 8u + (uint32_t)w8) * (uint32_t)Nk;
      if (DROP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] *= drop_mult(a.drop, SITE_ATTN_RG2KG, ibase + (uint32_t)acc_row(i, h));
      }
      const bf16x8 pf = as_frag(u32x4{pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])});
      *reinterpret_cast<u32x4*>(scr + 16 * lane) = vkg[s];        // [16 keys][64 bytes], linear
      const char* vp = scr + (4 * h + q4) * 64 + 2 * (16 * g1 + 4 * p4);
      const bf16x8 vf = join(lds_tr16(vp), lds_tr16(vp + 8 * 64));
      const f32x16 O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, zero16(), 0, 0, 0);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<u32x2*>(strip + (32 * s + l31) * PS + 2 * (8 * gq + 4 * h)) =
            u32x2{pack2(O[4 * gq], O[4 * gq + 1]), pack2(O[4 * gq + 2], O[4 * gq + 3])};
    }
  }
  // ---- arrival, second half: the block's barrier behind every wave's drain, then one relaxed ticket per sample
  stamp(a.stamps, 7);
  __syncthreads();                                                // strips complete; partial stores acknowledged; in-projection biases read
  if (tid < 320) *reinterpret_cast<float4*>(cst + 4 * tid) = c1v;
  int tk[RT];
  if (tid == 0) {
    int prev = -1;
#pragma unroll
    for (int s = 0; s < RT; ++s) {
      tk[s] = -1;
      if (sub[s].nr > 0 && sub[s].b != prev) {
        prev = sub[s].b;
        tk[s] = __hip_atomic_fetch_add(a.tickets + prev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  __syncthreads();                                                // the chain's constants are in LDS (and the partials' Z scratch is free for the row totals)
  stamp(a.stamps, 8);
  chain_tiles<RT, DEPTH, DROP, SAVE, false, true>(a, S, sub, strips, bufR, red, cst, w8, lane, sto);
  stamp(a.stamps, 12);
  if (tid == 0) {
#pragma unroll
    for (int s = 0; s < RT; ++s) {
      const int nseg = (toff[s][1] - 1) / RT - toff[s][0] / RT + 1;       // blocks the sample's tiles are spread over
      flag[s] = (tk[s] >= 0 && tk[s] == nseg - 1) ? nseg : -1;
    }
  }
  __syncthreads();
  int todo[RT];
#pragma unroll
  for (int s = 0; s < RT; ++s) todo[s] = flag[s];
  bool any = false;
#pragma unroll
  for (int s = 0; s < RT; ++s) any = any || todo[s] > 0;
  if (!any) return;                                               // (block-uniform)
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const BackStream& K = a.s[1];
  char* kbufO = strips; char* kscr = strips + 32 * PR; char* kbufY = bufR;
#pragma unroll 1
  for (int s = 0; s < RT; ++s) {
    const int nseg = __builtin_amdgcn_readfirstlane(todo[s]);
    if (nseg <= 0) continue;
    const int b = sub[s].b;
    __syncthreads();                                              // (the acquire above / the previous sample's chain is done with the buffers)
    for (int c = tid; c < 32 * PR / 16; c += NTH) reinterpret_cast<u32x4*>(kbufO)[c] = u32x4{0u, 0u, 0u, 0u};   // rows >= Nk stay zero
    StageW<1, 1, 16, 2, DEPTH> stk;
    stk.prefetch(reinterpret_cast<const u32x4*>(K.Wo) + (size_t)((w8 >> 1) * (16 * 2) + (w8 & 1)) * 64 + lane);
    load_consts(K, cst);
    __syncthreads();
    kg_combine<RT, SAVE>(a, kscr, kbufO, b, a.tile_off[b], nseg);
    __syncthreads();
    stamp(a.stamps, 13);
    const Sub ksub[1] = {Sub{b, b * Nk, Nk, 1.0f / (float)Nk}};
    chain_tiles<1, DEPTH, DROP, SAVE, false>(a, K, ksub, kbufO, kbufY, red, cst, w8, lane, stk);
    stamp(a.stamps, 14);
  }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int RT>
int front8_launch(FrontArgs& a, int total, hipStream_t stream) {
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&front8_kernel<RT, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, FrontCfg<RT>::LDS);
    return true;
  }();
  (void)attr;
  hipLaunchKernelGGL((front8_kernel<RT, 12>), dim3(total), dim3(NTH), FrontCfg<RT>::LDS, stream, a);
  return 0;
}
template <int RT, bool DROP, bool SAVE>
int back8_launch2(BackArgs& a, hipStream_t stream) {
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&back8_kernel<RT, 8, DROP, SAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, BackCfg<RT>::LDS);
    return true;
  }();
  (void)attr;
  hipLaunchKernelGGL((back8_kernel<RT, 8, DROP, SAVE>), dim3((a.rg_tiles_max + RT - 1) / RT), dim3(NTH), BackCfg<RT>::LDS, stream, a);
  return 0;
}
// dropout and the saved-for-backward set are compile-time variants: as run-time flags they were two branches per ELEMENT of every epilogue
template <int RT>
int back8_launch(BackArgs& a, hipStream_t stream) {
  const bool drop = a.drop.p > 0.f, save = a.save != 0;
  if (drop) return save ? back8_launch2<RT, true, true>(a, stream) : back8_launch2<RT, true, false>(a, stream);
  return save ? back8_launch2<RT, false, true>(a, stream) : back8_launch2<RT, false, false>(a, stream);
}

template <int RT, bool DROP, bool SAVE>
int rgfwd_launch2(RgFwdArgs& g, hipStream_t stream) {
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rgfwd_kernel<RT, DROP, SAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, RgCfg<RT>::LDS);
    return true;
  }();
  (void)attr;
  hipLaunchKernelGGL((rgfwd_kernel<RT, DROP, SAVE>), dim3((g.b.rg_tiles_max + RT - 1) / RT), dim3(NTH), RgCfg<RT>::LDS, stream, g);
  return 0;
}
template <int RT>
int rgfwd_launch(RgFwdArgs& g, hipStream_t stream) {
  const bool drop = g.b.drop.p > 0.f, save = g.b.save != 0;
  if (drop) return save ? rgfwd_launch2<RT, true, true>(g, stream) : rgfwd_launch2<RT, true, false>(g, stream);
  return save ? rgfwd_launch2<RT, false, true>(g, stream) : rgfwd_launch2<RT, false, false>(g, stream);
}

}  // namespace

int launch_wide_front(FrontArgs& a, int rt, hipStream_t stream, int kg_only) {
  if (rt != 1 && rt != 2 && rt != 4) return (int)hipErrorInvalidValue;
  int total = 0;
  for (int i = 0; i < 2; ++i) {
    FrontStream& S = a.s[i];
    if (kg_only && i == 0) { S.tile_begin = 0; continue; }      // (every block then belongs to stream 1)
    if (S.M < 1 || !S.X || !S.W0 || !S.W1 || !S.b0 || !S.bq || !S.bkv || !S.R16 || !S.Q16 || !S.KV16 || (a.save && !S.X16))
      return (int)hipErrorInvalidValue;
    if (!al16(S.X) || !al16(S.b0) || !al16(S.bq) || !al16(S.bkv) || !al16(S.R16) || !al16(S.Q16) || !al16(S.KV16) || !al16(S.W0) || !al16(S.W1) || (a.save && !al16(S.X16)))
      return (int)hipErrorInvalidValue;
    S.tile_begin = total;
    total += (S.M + 32 * rt - 1) / (32 * rt);
  }
  if (a.nzero < 0 || a.nzero > FUSED_FRONT_MAXZ) return (int)hipErrorInvalidValue;
  for (int i = 0; i < a.nzero; ++i)
    if (!a.zero_ptr[i] || !al16(a.zero_ptr[i]) || (a.zero_bytes[i] & 15)) return (int)hipErrorInvalidValue;
  if (a.nxjob < 0 || a.nxjob > FUSED_FRONT_MAXX) return (int)hipErrorInvalidValue;
  a.xchunks = 0;
  for (int i = 0; i < a.nxjob; ++i) {
    ShadowJob& J = a.xjob[i];
    if ((J.N & 127) || (J.K & 15) || J.nsrc < 1 || J.nsrc > 4 || !J.dst || !al16(J.dst)) return (int)hipErrorInvalidValue;
    int sum = 0;
    for (int q = 0; q < J.nsrc; ++q) { if (!J.src[q] || !al16(J.src[q]) || (J.ld[q] & 3) || (J.transposed && (J.rows[q] & 7))) return (int)hipErrorInvalidValue; sum += J.rows[q]; }
    if (sum != (J.transposed ? J.K : J.N)) return (int)hipErrorInvalidValue;      // (a transposed job's sources stack along K)
    J.chunk_begin = a.xchunks;
    a.xchunks += J.N * J.K / 8;
  }
  if (a.split3) total *= 3;
  a.xblock0 = total;
  total += (a.xchunks + NTH - 1) / NTH;
  const int prof = gemm_prof_open(stream, 2.0 * ((kg_only ? 0.0 : (double)a.s[0].M) + a.s[1].M) * (128.0 * 256.0 + 256.0 * 768.0), PROF_FRONT);
  if (rt == 1) front8_launch<1>(a, total, stream); else if (rt == 2) front8_launch<2>(a, total, stream); else front8_launch<4>(a, total, stream);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

int wide_max_rows(int rt) { return 32 * rt * FUSED_WIDE_MAXSEG; }   // (segments of a sample <= FUSED_WIDE_MAXSEG: the combine's LDS tables)

int launch_wide_back(BackArgs& a, int rt, int max_nr, hipStream_t stream) {
  if (rt != 1 && rt != 2 && rt != 4) return (int)hipErrorInvalidValue;
  if (a.B < 1 || a.Nk < 1 || a.Nk > 16 || a.rg_tiles_max < 1 || !a.Q16 || !a.KV16 || !a.Q2_16 || !a.KV2_16 || !a.off || !a.tile_off || !a.tile_desc || !a.inv_nr ||
      !a.part || !a.tickets)
    return (int)hipErrorInvalidValue;
  if ((max_nr + 31) / 32 / rt + 2 > FUSED_WIDE_MAXSEG) return (int)hipErrorInvalidValue;
  for (int i = 0; i < 2; ++i) {
    const BackStream& S = a.s[i];
    if (!S.Wo || !S.bo || !S.W1 || !S.b1 || !S.ln_g || !S.ln_b || !S.R16 || !S.Ymean || !S.Hmean) return (int)hipErrorInvalidValue;
    if (a.save && (!S.O16 || !S.Y16 || !S.XH16 || !S.rstd || !S.mask)) return (int)hipErrorInvalidValue;
    if (!al16(S.bo) || !al16(S.ln_g) || !al16(S.ln_b) || !al16(S.Wo) || !al16(S.W1) || !al16(S.R16)) return (int)hipErrorInvalidValue;
  }
  const double rows = (double)a.rows_rg + (double)a.B * a.Nk;
  const int prof = gemm_prof_open(stream, 2.0 * rows * (256.0 * 256.0 + 256.0 * 512.0) + 8.0 * (double)a.rows_rg * a.Nk * 256.0, PROF_BACK);
  if (rt == 1) back8_launch<1>(a, stream); else if (rt == 2) back8_launch<2>(a, stream); else back8_launch<4>(a, stream);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

// The RG rows' whole forward in one launch (rgfwd_kernel).  `f` = the RG stream of the front half (X, shadows, biases, the
// saved operands X16 / R16 / Q16 / KV16 = the RG keys|values), `b` = the back half's arguments.  The KG rows' projections
// (b.Q2_16, b.KV16) must already exist: launch_wide_front(..., kg_only = 1) first.
int launch_wide_rgfwd(const FrontStream& f, float qscale, BackArgs& b, int rt, int max_nr, hipStream_t stream) {
  if (rt != 2 && rt != 4) return (int)hipErrorInvalidValue;
  if (b.B < 1 || b.Nk < 1 || b.Nk > 16 || b.rg_tiles_max < 1 || !b.KV16 || !b.Q2_16 || !b.off || !b.tile_off || !b.tile_desc || !b.inv_nr || !b.part || !b.tickets)
    return (int)hipErrorInvalidValue;
  if ((max_nr + 31) / 32 / rt + 2 > FUSED_WIDE_MAXSEG) return (int)hipErrorInvalidValue;
  if (!f.X || !f.W0 || !f.W1 || !f.b0 || !f.bq || !f.bkv || !al16(f.X) || !al16(f.b0) || !al16(f.bq) || !al16(f.bkv) || !al16(f.W0) || !al16(f.W1)) return (int)hipErrorInvalidValue;
  if (b.save && (!f.X16 || !f.R16 || !f.Q16 || !f.KV16 || !al16(f.X16))) return (int)hipErrorInvalidValue;
  for (int i = 0; i < 2; ++i) {
    const BackStream& S = b.s[i];
    if (!S.Wo || !S.bo || !S.W1 || !S.b1 || !S.ln_g || !S.ln_b || !S.Ymean || !S.Hmean || (i == 1 && !S.R16)) return (int)hipErrorInvalidValue;
    if (b.save && (!S.O16 || !S.Y16 || !S.XH16 || !S.rstd || !S.mask || !S.R16)) return (int)hipErrorInvalidValue;
    if (!al16(S.bo) || !al16(S.ln_g) || !al16(S.ln_b) || !al16(S.b1) || !al16(S.Wo) || !al16(S.W1)) return (int)hipErrorInvalidValue;
  }
  RgFwdArgs g; g.f = f; g.b = b; g.qscale = qscale;
  // executed FLOPs per RG row: 128 -> 256, 256 -> 768, 256 -> 256, 256 -> 512 and both attention directions; the KG rows' chain on top
  const double rows = (double)b.rows_rg, kgrows = (double)b.B * b.Nk;
  const int prof = gemm_prof_open(stream, 2.0 * rows * (128.0 * 256.0 + 256.0 * 768.0 + 256.0 * 256.0 + 256.0 * 512.0) + 8.0 * rows * b.Nk * 256.0 +
                                  2.0 * kgrows * (256.0 * 256.0 + 256.0 * 512.0), PROF_BACK);
  if (rt == 2) rgfwd_launch<2>(g, stream); else rgfwd_launch<4>(g, stream);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}
