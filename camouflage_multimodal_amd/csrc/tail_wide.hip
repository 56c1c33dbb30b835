// Per-sample tail for large batches (gfx950); contract in tail_wide.h.
//
// Why.  Behind the node-level kernels every sample is ONE row through five small dense layers (1.2 MFLOP).  As one GEMM launch
// per layer that is five launches whose M = B rows fill 8 CUs: 24 us each at B = 256, a third of the whole forward.  Here a
// block keeps 32 samples on chip through all five layers and streams the weights (2.4 MB as bf16 hi + lo planes) from L2 in MFMA
// fragment order, as the row-tile kernels do: ~25 us whatever B is, B / 32 blocks.
//
// Precision.  The reference computes these layers in fp32 and the logits come straight out of them, so operands are NOT rounded to
// bf16 here: x = x_hi + x_lo, W = W_hi + W_lo (both bf16) and x W ~ x_hi W_hi + x_lo W_hi + x_hi W_lo on the bf16 MFMA with fp32
// accumulation -- the dropped x_lo W_lo term is 2^-18 relative.
#include "tail_wide.h"
#include "gemm.h"      // launch timing hooks

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ float bf2f(uint32_t b) { return __uint_as_float(b << 16); }
// two values -> their packed hi and lo planes
struct HiLo { uint32_t hi, lo; };
__device__ __forceinline__ HiLo split2(float a, float b) {
  const uint32_t ha = f2bf(a), hb = f2bf(b);
  return HiLo{ha | (hb << 16), pack2(a - bf2f(ha), b - bf2f(hb))};
}

constexpr int NW = 8, NTH = 64 * NW, TS = 32;
constexpr int P512 = 1040, P256 = 528;              // row pitches (bytes) of a [32][512] / [32][256] bf16 plane
constexpr int PLANE = TS * P512;                    // 33 280
constexpr int BUF = 2 * PLANE;                      // hi | lo
constexpr int PHID = 2064;                          // fp32 [32][512] pitch (bytes): 32 * 2064 <= BUF
static_assert(TS * PHID <= BUF, "the hidden tile fits a plane pair");
constexpr int C_B13 = 0, C_B23 = 256, C_BFU0 = 512, C_BFU3 = 768, C_BH0 = 1024, C_BH3 = 1536, C_WH3 = 1536 + 32;      // floats
constexpr int CONSTS = C_WH3 + (2 * TAILW_MAXC + 2) * 128;
constexpr int LDS_BYTES = 2 * BUF + CONSTS * 4;

// out^T tile(s) of one layer: acc[t] += W_hi x_hi + W_hi x_lo + W_lo x_hi over KS k steps; the wave's hi / lo fragment streams
// start at wh / wl (+ lane; 16-byte units), fragment (ks, t) at [64 (ks KST + t)]; x planes in LDS at xh / xl (this lane's first
// fragment: row lane & 31, byte 16 (lane >> 5)), k step ks 32 bytes further.
template <int NT, int KS, int KST, int DEPTH>
struct StageHL {
  static constexpr int TOTAL = NT * KS;
  static constexpr int D = DEPTH < TOTAL ? DEPTH : TOTAL;
  u32x4 bh[D], bl[D];
  const u32x4 *wh, *wl;
  __device__ __forceinline__ int off(int i) const { return 64 * ((i / NT) * KST + (i % NT)); }
  __device__ __forceinline__ void prefetch(const u32x4* __restrict__ wh_, const u32x4* __restrict__ wl_) {
    wh = wh_; wl = wl_;
#pragma unroll
    for (int i = 0; i < D; ++i) { bh[i] = wh[off(i)]; bl[i] = wl[off(i)]; }
  }
  __device__ __forceinline__ void run(const char* xh, const char* xl, f32x16 (&acc)[NT]) {
    bf16x8 xhi, xlo, nhi, nlo;
    nhi = *reinterpret_cast<const bf16x8*>(xh); nlo = *reinterpret_cast<const bf16x8*>(xl);
#pragma unroll
    for (int i = 0; i < TOTAL; ++i) {
      const int ks = i / NT, t = i % NT;
      const bf16x8 whf = as_frag(bh[i % D]), wlf = as_frag(bl[i % D]);
      if (i + D < TOTAL) { bh[i % D] = wh[off(i + D)]; bl[i % D] = wl[off(i + D)]; }
      if (t == 0) {
        xhi = nhi; xlo = nlo;
        if (ks + 1 < KS) { nhi = *reinterpret_cast<const bf16x8*>(xh + 32 * (ks + 1)); nlo = *reinterpret_cast<const bf16x8*>(xl + 32 * (ks + 1)); }
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whf, xhi, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whf, xlo, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlf, xhi, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// rows b0 .. b0 + 31 (clamped to B - 1) of a pooled fp32 tensor [B][512] -> hi / lo planes of a plane pair
__device__ __forceinline__ void load_in512(const float* src, int b0, int B, float4 (&v)[8]) {
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = threadIdx.x + NTH * it, r = idx >> 7, c4 = idx & 127;
    v[it] = *reinterpret_cast<const float4*>(src + (size_t)min(b0 + r, B - 1) * 512 + 4 * c4);
  }
}
__device__ __forceinline__ void store_in512(char* buf, const float4 (&v)[8]) {
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = threadIdx.x + NTH * it, r = idx >> 7, c4 = idx & 127;
    const HiLo p = split2(v[it].x, v[it].y), q = split2(v[it].z, v[it].w);
    *reinterpret_cast<u32x2*>(buf + r * P512 + 8 * c4) = u32x2{p.hi, q.hi};
    *reinterpret_cast<u32x2*>(buf + PLANE + r * P512 + 8 * c4) = u32x2{p.lo, q.lo};
  }
}

template <bool DROP>
__global__ __launch_bounds__(NTH, 2) void tailw_fwd_kernel(const TailWideArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = (int)blockIdx.x * TS, B = a.B, C = a.C, Wd = 2 * C + 2;
  char* bufA = smem; char* bufB = smem + BUF;
  float* cst = reinterpret_cast<float*>(smem + 2 * BUF);
  const size_t frag0 = (size_t)((w8 >> 1) * 2 * 32 + (w8 & 1)) * 64 + lane;       // [4][32 k steps][2 tiles]: tile w8 of a [256 x 512] shadow
  StageHL<1, 32, 2, 6> s1;
  s1.prefetch(reinterpret_cast<const u32x4*>(a.T13h) + frag0, reinterpret_cast<const u32x4*>(a.T13l) + frag0);
  float4 in[8];
  load_in512(a.H1mean, b0, B, in);
  // biases and the head output layers -> LDS (all loads of a thread in flight together)
  {
    constexpr int CIT = (CONSTS + NTH - 1) / NTH;
    float cv[CIT];
#pragma unroll
    for (int it = 0; it < CIT; ++it) {
      const int i = tid + NTH * it;
      const float* p = nullptr;
      if (i < C_B23) p = a.b13 + i;
      else if (i < C_BFU0) p = a.b23 + (i - C_B23);
      else if (i < C_BFU3) p = a.bfu0 + (i - C_BFU0);
      else if (i < C_BH0) p = a.bfu3 + (i - C_BFU3);
      else if (i < C_BH3) p = a.bh0[(i - C_BH0) >> 7] + ((i - C_BH0) & 127);
      else if (i < CONSTS) {                                                     // output o of the 2C+2: head x, its row r
        const bool isw = i >= C_WH3;
        const int o = isw ? (i - C_WH3) >> 7 : i - C_BH3, k = (i - C_WH3) & 127;
        if (o < Wd) {
          const int x = o < C ? 0 : (o < 2 * C ? 1 : (o == 2 * C ? 2 : 3));
          const int r = o - (x == 0 ? 0 : (x == 1 ? C : (x == 2 ? 2 * C : 2 * C + 1)));
          p = isw ? a.Wh3[x] + (size_t)r * 128 + k : a.bh3[x] + r;
        }
      }
      cv[it] = p ? *p : 0.f;
    }
#pragma unroll
    for (int it = 0; it < CIT; ++it) { const int i = tid + NTH * it; if (i < CONSTS) cst[i] = cv[it]; }
  }
  store_in512(bufA, in);
  load_in512(a.H2mean, b0, B, in);                           // (lands while the first layer runs)
  __syncthreads();
  const char* xrowA = bufA + l31 * P512 + 16 * h;
  const size_t srow = (size_t)min(b0 + l31, B - 1);
  // epilogue helper: this lane's 16 values of tile column block `col0` (+ acc_row) -> hi / lo planes of `dst` (pitch `pitch`)
  auto put_hl = [&](char* dst, int pitch, int plane, int col0, const float (&v)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const HiLo p = split2(v[4 * g], v[4 * g + 1]), q = split2(v[4 * g + 2], v[4 * g + 3]);
      *reinterpret_cast<u32x2*>(dst + l31 * pitch + 2 * (col0 + 8 * g + 4 * h)) = u32x2{p.hi, q.hi};
      *reinterpret_cast<u32x2*>(dst + plane + l31 * pitch + 2 * (col0 + 8 * g + 4 * h)) = u32x2{p.lo, q.lo};
    }
  };
  // ---- layer 1: comb = [Ymean + W13 H1mean + b13 | Y2mean + W23 H2mean + b23]  (fusion_model.py:120,131,134-136; the pooled second
  // FFN layers).  Feature tile w8 of each half.
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const float* res = (half ? a.Y2mean : a.Ymean) + srow * 256 + 32 * w8 + 4 * h;
    float4 rv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) rv[g] = *reinterpret_cast<const float4*>(res + 8 * g);
    f32x16 acc[1] = {zero16()};
    s1.run(xrowA, xrowA + PLANE, acc);
    if (half == 0) s1.prefetch(reinterpret_cast<const u32x4*>(a.T23h) + frag0, reinterpret_cast<const u32x4*>(a.T23l) + frag0);
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float* bb = cst + (half ? C_B23 : C_B13) + 32 * w8 + 8 * g + 4 * h;
      v[4 * g] = acc[0][4 * g] + bb[0] + rv[g].x; v[4 * g + 1] = acc[0][4 * g + 1] + bb[1] + rv[g].y;
      v[4 * g + 2] = acc[0][4 * g + 2] + bb[2] + rv[g].z; v[4 * g + 3] = acc[0][4 * g + 3] + bb[3] + rv[g].w;
    }
    put_hl(bufB, P512, PLANE, 256 * half + 32 * w8, v);
    if (a.comb_out && b0 + l31 < B) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(a.comb_out + (size_t)(b0 + l31) * 512 + 256 * half + 32 * w8 + 8 * g + 4 * h) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
    }
    if (half == 0) {
      __syncthreads();                                         // every wave is done with the first input
      store_in512(bufA, in);
      __syncthreads();
    }
  }
  StageHL<1, 32, 2, 6> s2;
  s2.prefetch(reinterpret_cast<const u32x4*>(a.Tfu0h) + frag0, reinterpret_cast<const u32x4*>(a.Tfu0l) + frag0);
  __syncthreads();                                             // comb complete
  // ---- layer 2: F1 = dropout(relu(Wfu0 comb + bfu0))  (fusion_model.py:68-71,138)
  StageHL<1, 16, 2, 6> s3;
  const size_t frag3 = (size_t)((w8 >> 1) * 2 * 16 + (w8 & 1)) * 64 + lane;        // [4][16][2]: tile w8 of a [256 x 256] shadow
  {
    f32x16 acc[1] = {zero16()};
    s2.run(bufB + l31 * P512 + 16 * h, bufB + PLANE + l31 * P512 + 16 * h, acc);
    s3.prefetch(reinterpret_cast<const u32x4*>(a.Tfu3h) + frag3, reinterpret_cast<const u32x4*>(a.Tfu3l) + frag3);
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int col = 32 * w8 + acc_row(i, h);
      v[i] = fmaxf(acc[0][i] + cst[C_BFU0 + col], 0.f);
      if (DROP) v[i] *= drop_mult(a.drop, SITE_FUSE, (uint32_t)(b0 + l31) * 256u + (uint32_t)col);
    }
    put_hl(bufA, P256, TS * P256, 32 * w8, v);
    if (a.F1_out && b0 + l31 < B) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(a.F1_out + (size_t)(b0 + l31) * 256 + 32 * w8 + 8 * g + 4 * h) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
    }
  }
  __syncthreads();
  // ---- layer 3: fused = Wfu3 F1 + bfu3  (fusion_model.py:72)
  StageHL<2, 16, 4, 6> s4;
  const size_t frag4 = (size_t)((w8 >> 1) * 4 * 16 + 2 * (w8 & 1)) * 64 + lane;    // [4][16][4]: tiles 2 w8, 2 w8 + 1 of the stacked [512 x 256] head layers
  {
    f32x16 acc[1] = {zero16()};
    s3.run(bufA + l31 * P256 + 16 * h, bufA + TS * P256 + l31 * P256 + 16 * h, acc);
    s4.prefetch(reinterpret_cast<const u32x4*>(a.Th0h) + frag4, reinterpret_cast<const u32x4*>(a.Th0l) + frag4);
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = acc[0][i] + cst[C_BFU3 + 32 * w8 + acc_row(i, h)];
    put_hl(bufB, P256, TS * P256, 32 * w8, v);
    if (a.fused_out && b0 + l31 < B) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(a.fused_out + (size_t)(b0 + l31) * 256 + 32 * w8 + 8 * g + 4 * h) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
    }
  }
  __syncthreads();
  // ---- layer 4: the four heads' hidden layers, hid_x = dropout(relu(Wh0_x fused + bh0_x)), x = column / 128  (fusion_model.py:208-235)
  {
    f32x16 acc[2] = {zero16(), zero16()};
    s4.run(bufB + l31 * P256 + 16 * h, bufB + TS * P256 + l31 * P256 + 16 * h, acc);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 o;
        float* op = &o.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = 32 * (2 * w8 + t) + 8 * g + 4 * h + j;
          float v = fmaxf(acc[t][4 * g + j] + cst[C_BH0 + col], 0.f);
          if (DROP) v *= drop_mult(a.drop, SITE_HEAD0 + (uint32_t)(col >> 7), (uint32_t)(b0 + l31) * 128u + (uint32_t)(col & 127));
          op[j] = v;
        }
        *reinterpret_cast<float4*>(bufA + l31 * PHID + 4 * (32 * (2 * w8 + t) + 8 * g + 4 * h)) = o;
        if (a.hid_out && b0 + l31 < B) *reinterpret_cast<float4*>(a.hid_out + (size_t)(b0 + l31) * 512 + 32 * (2 * w8 + t) + 8 * g + 4 * h) = o;
      }
  }
  if (a.hid_out) return;                                     // (training call: the head output layer belongs to the loss launch)
  __syncthreads();
  // ---- layer 5: the head outputs (fp32, VALU): 16 lanes per sample, 8 hidden units each; score head through the sigmoid
  {
    const int s = tid >> 4, sub = tid & 15;
    for (int o = 0; o < Wd; ++o) {
      const int x = o < C ? 0 : (o < 2 * C ? 1 : (o == 2 * C ? 2 : 3));
      const float4* hp = reinterpret_cast<const float4*>(bufA + s * PHID + 4 * (128 * x + 8 * sub));
      const float4* wp = reinterpret_cast<const float4*>(cst + C_WH3 + 128 * o + 8 * sub);
      const float4 h0 = hp[0], h1 = hp[1], w0 = wp[0], w1 = wp[1];
      float d = h0.x * w0.x + h0.y * w0.y + h0.z * w0.z + h0.w * w0.w + h1.x * w1.x + h1.y * w1.y + h1.z * w1.z + h1.w * w1.w;
      d += __shfl_xor(d, 8, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 1, 64);
      d += cst[C_BH3 + o];
      if (o == 2 * C + 1) d = 1.0f / (1.0f + __expf(-d));
      if (sub == 0 && b0 + s < B) a.outs[(size_t)(b0 + s) * Wd + o] = d;
    }
  }
}

// ---------------------------------------------------------------- the input-gradient chain (tail_wide.h, TailWideBwdArgs)
// Same machinery as the forward: 32 samples per block, 8 waves, every product three bf16 MFMAs on hi / lo planes; the planes
// ping-pong between the two LDS buffers.  Every result also leaves as fp32 (the operands of the weight-gradient launch, and the
// pooled gradients the node-level backward reads).
__global__ __launch_bounds__(NTH, 2) void tailw_bwd_kernel(const TailWideBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = (int)blockIdx.x * TS, B = a.B;
  char* bufA = smem; char* bufB = smem + BUF;
  const size_t frag0 = (size_t)((w8 >> 1) * 2 * 32 + (w8 & 1)) * 64 + lane;       // tile w8 of a [256 x 512] shadow
  const size_t frag3 = (size_t)((w8 >> 1) * 2 * 16 + (w8 & 1)) * 64 + lane;       // tile w8 of a [256 x 256] shadow
  const size_t frag4 = (size_t)((w8 >> 1) * 4 * 16 + 2 * (w8 & 1)) * 64 + lane;   // tiles 2 w8, 2 w8 + 1 of a [512 x 256] shadow
  StageHL<1, 32, 2, 6> s1;
  s1.prefetch(reinterpret_cast<const u32x4*>(a.Th0h) + frag0, reinterpret_cast<const u32x4*>(a.Th0l) + frag0);
  float4 in[8];
  load_in512(a.dhid, b0, B, in);
  const bool rok = b0 + l31 < B;
  const size_t srow = (size_t)min(b0 + l31, B - 1);
  float4 f1v[4];                                             // this lane's 16 F1 values of tile w8 (the ReLU mask of layer 2)
#pragma unroll
  for (int g = 0; g < 4; ++g) f1v[g] = *reinterpret_cast<const float4*>(a.F1 + srow * 256 + 32 * w8 + 8 * g + 4 * h);
  store_in512(bufA, in);
  __syncthreads();
  auto put_hl = [&](char* dst, int pitch, int plane, int col0, const float (&v)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const HiLo p = split2(v[4 * g], v[4 * g + 1]), q = split2(v[4 * g + 2], v[4 * g + 3]);
      *reinterpret_cast<u32x2*>(dst + l31 * pitch + 2 * (col0 + 8 * g + 4 * h)) = u32x2{p.hi, q.hi};
      *reinterpret_cast<u32x2*>(dst + plane + l31 * pitch + 2 * (col0 + 8 * g + 4 * h)) = u32x2{p.lo, q.lo};
    }
  };
  auto put_g = [&](float* out, int ld, int col0, const float (&v)[16]) {
    if (!rok) return;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(out + (size_t)(b0 + l31) * ld + col0 + 8 * g + 4 * h) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
  };
  StageHL<1, 16, 2, 6> s2;
  // ---- dfused = dhid . Wh0s  (K = 512)
  {
    f32x16 acc[1] = {zero16()};
    s1.run(bufA + l31 * P512 + 16 * h, bufA + PLANE + l31 * P512 + 16 * h, acc);
    s2.prefetch(reinterpret_cast<const u32x4*>(a.Tfu3h) + frag3, reinterpret_cast<const u32x4*>(a.Tfu3l) + frag3);
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = acc[0][i];
    put_hl(bufB, P256, TS * P256, 32 * w8, v);
    put_g(a.dfused, 256, 32 * w8, v);
  }
  __syncthreads();
  StageHL<2, 16, 4, 6> s3;
  // ---- dF1 = (F1 > 0) * scale * (dfused . Wfu3)  (K = 256)
  {
    f32x16 acc[1] = {zero16()};
    s2.run(bufB + l31 * P256 + 16 * h, bufB + TS * P256 + l31 * P256 + 16 * h, acc);
    s3.prefetch(reinterpret_cast<const u32x4*>(a.Tfu0h) + frag4, reinterpret_cast<const u32x4*>(a.Tfu0l) + frag4);
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float m[4] = {f1v[g].x, f1v[g].y, f1v[g].z, f1v[g].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * g + e] = m[e] > 0.f ? acc[0][4 * g + e] * a.scale : 0.f;
    }
    put_hl(bufA, P256, TS * P256, 32 * w8, v);
    put_g(a.dF1, 256, 32 * w8, v);
  }
  __syncthreads();
  StageHL<2, 16, 4, 6> s4;
  // ---- dcomb = dF1 . Wfu0  (K = 256, 512 outputs: tiles 2 w8, 2 w8 + 1)
  {
    f32x16 acc[2] = {zero16(), zero16()};
    s3.run(bufA + l31 * P256 + 16 * h, bufA + TS * P256 + l31 * P256 + 16 * h, acc);
    s4.prefetch(reinterpret_cast<const u32x4*>(a.T13h) + frag4, reinterpret_cast<const u32x4*>(a.T13l) + frag4);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = acc[t][i];
      put_hl(bufB, P512, PLANE, 32 * (2 * w8 + t), v);
      put_g(a.dcomb, 512, 32 * (2 * w8 + t), v);
    }
  }
  __syncthreads();
  // ---- d(mean H1d) = dcomb[:, :256] . W13,  d(mean H2d) = dcomb[:, 256:] . W23  (K = 256 each, 512 outputs)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x16 acc[2] = {zero16(), zero16()};
    s4.run(bufB + l31 * P512 + 512 * half + 16 * h, bufB + PLANE + l31 * P512 + 512 * half + 16 * h, acc);
    if (half == 0) s4.prefetch(reinterpret_cast<const u32x4*>(a.T23h) + frag4, reinterpret_cast<const u32x4*>(a.T23l) + frag4);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = acc[t][i];
      put_g(half ? a.dHm2 : a.dHm1, 512, 32 * (2 * w8 + t), v);
    }
  }
}

}  // namespace

int tail_wide_ok(int B, int C) { return B >= 1 && C >= 1 && C <= TAILW_MAXC; }

int launch_tail_wide(const TailWideArgs& a, hipStream_t stream) {
  const bool saves = a.comb_out || a.F1_out || a.fused_out || a.hid_out;
  if (saves && !(a.comb_out && a.F1_out && a.fused_out && a.hid_out)) return (int)hipErrorInvalidValue;
  if (!tail_wide_ok(a.B, a.C) || !a.Ymean || !a.H1mean || !a.Y2mean || !a.H2mean || (!a.outs && !saves) || !a.T13h || !a.T13l || !a.T23h || !a.T23l ||
      !a.Tfu0h || !a.Tfu0l || !a.Tfu3h || !a.Tfu3l || !a.Th0h || !a.Th0l || !a.b13 || !a.b23 || !a.bfu0 || !a.bfu3)
    return (int)hipErrorInvalidValue;
  for (int x = 0; x < 4; ++x) if (!a.bh0[x] || !a.Wh3[x] || !a.bh3[x]) return (int)hipErrorInvalidValue;
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tailw_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tailw_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    return true;
  }();
  (void)attr;
  const int prof = gemm_prof_open(stream, 2.0 * a.B * (2.0 * 512 * 256 + 512 * 256 + 256 * 256 + 256 * 512) * 3.0, PROF_TAIL);
  const dim3 grid((a.B + TS - 1) / TS);
  if (a.drop.p > 0.f) hipLaunchKernelGGL((tailw_fwd_kernel<true>), grid, dim3(NTH), LDS_BYTES, stream, a);
  else                hipLaunchKernelGGL((tailw_fwd_kernel<false>), grid, dim3(NTH), LDS_BYTES, stream, a);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

int launch_tail_wide_bwd(const TailWideBwdArgs& a, hipStream_t stream) {
  if (a.B < 1 || !a.dhid || !a.F1 || !a.Th0h || !a.Th0l || !a.Tfu3h || !a.Tfu3l || !a.Tfu0h || !a.Tfu0l || !a.T13h || !a.T13l || !a.T23h || !a.T23l ||
      !a.dfused || !a.dF1 || !a.dcomb || !a.dHm1 || !a.dHm2)
    return (int)hipErrorInvalidValue;
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tailw_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    return true;
  }();
  (void)attr;
  const int prof = gemm_prof_open(stream, 2.0 * a.B * (512.0 * 256 + 256.0 * 256 + 3.0 * 256 * 512) * 3.0, PROF_TAIL);
  hipLaunchKernelGGL(tailw_bwd_kernel, dim3((a.B + TS - 1) / TS), dim3(NTH), 2 * BUF, stream, a);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}
