// Grouped MFMA GEMM for gfx950 (MI355X): C = epi(A.B + bias) (+res), fp32 in HBM.
//
// Block tile 64(M) x 128(N) x 32(K), 256 threads = 4 waves as 2(M) x 2(N); each wave
// owns a 32 x 64 sub-tile = two 32x32 MFMA accumulators that share the A fragment.
//
//   PREC_F32 : v_mfma_f32_32x32x2_f32 -- exact fp32 FMA chains (parity mode, 157 TF peak)
//   PREC_BF16: operands rounded to bf16 while they are staged into LDS,
//              v_mfma_f32_32x32x16_bf16, fp32 accumulate (2.5 PF peak)
//
// Operand layouts are per-problem runtime flags so one launch can mix the three GEMM
// kinds of a training step:  y = x.W^T (A m-major, B n-major), dx = dy.W (B k-major),
// dW = dy^T.x (both k-major, split-K + atomic accumulate).
//
// LDS images (conflict-free by construction, see DESIGN.md "GEMM"):
//   f32  m-major  [rows][36]   : 4 x ds_read_b128 per lane per K-tile; 36*4 B rows => a 16-lane
//                                b128 group touches 16 distinct 16-B slots of the 256-B bank row
//   f32  k-major  [32][rows+4] : ds_read_b32, lanes 0-31 consecutive
//   bf16 both     [rows][40]   : (80-B rows) 2 x ds_read_b128 per lane per K-tile; k-major sources are
//                                transposed in registers (4x4) while staging
// The MFMA k-slot a lane half feeds is arbitrary as long as A and B agree: lane half h of
// step s takes k = 16h + s (f32) or the 8 k's [16*step + 8h, +8) (bf16).
#include "gemm.h"

#include <vector>

namespace {

constexpr int BM = 64, BN = 128, BK = 32;
constexpr int LDM = BK + 4;    // 36 floats
constexpr int LDKA = BM + 4;   // 68
constexpr int LDKB = BN + 4;   // 132
constexpr int A_F32 = (BM * LDM > BK * LDKA) ? BM * LDM : BK * LDKA;   // 2304 floats
constexpr int B_F32 = (BN * LDM > BK * LDKB) ? BN * LDM : BK * LDKB;   // 4608 floats
constexpr int LDH = BK + 8;    // bf16 row stride (elements): 40 -> 80 B
constexpr int A_BF16 = BM * LDH;   // elements (ushort)
constexpr int B_BF16 = BN * LDH;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float4 ld4(const float* __restrict__ p, int nvalid, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid >= 4 && vec) {
    v = *reinterpret_cast<const float4*>(p);
  } else {
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
  }
  return v;
}

__device__ __forceinline__ int clamp4(int n) { return n < 0 ? 0 : (n > 4 ? 4 : n); }

__device__ __forceinline__ unsigned short f2bf(float f) {
  // round-to-nearest-even; NaN stays NaN through the compiler's own cast path
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

template <int PREC>
__global__ __launch_bounds__(256) void gemm_grouped_kernel(const GemmBatch gb, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];

  // ---- XCD-aware block remap (bijective): blocks that share an XCD (bid % 8) get a
  // contiguous run of tiles, so neighbouring N-tiles re-use the A row panel from that L2.
  int bid = blockIdx.x;
  {
    const int nwg = total_tiles, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAXP; ++i)
    if (i < gb.n && bid >= gb.p[i].tile_begin) pi = i;
  const GemmProb& P = gb.p[pi];

  int t = bid - P.tile_begin;
  const int ks = t % P.ksplit; t /= P.ksplit;
  const int tn = t % P.tiles_n;
  const int tm = t / P.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = ks * P.kchunk;
  const int kend = min(P.K, kbeg + P.kchunk);
  const int M = P.M, N = P.N;
  const bool akm = P.flags & GF_A_KMAJOR, bkm = P.flags & GF_B_KMAJOR;
  const bool vecA = ((P.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.A) & 15) == 0);
  const bool vecB = ((P.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.B) & 15) == 0);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  float4 ra[2], rb[4];

  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int s = tid + 256 * i;
      if (!akm) {
        const int r = s >> 3, c4 = s & 7;
        const int row = m0 + r, k = k0 + c4 * 4;
        const int nv = row < M ? clamp4(kend - k) : 0;
        ra[i] = ld4(P.A + (size_t)row * P.lda + k, nv, vecA);
      } else {
        const int kk = s >> 4, c4 = s & 15;
        const int k = k0 + kk, m = m0 + c4 * 4;
        const int nv = k < kend ? clamp4(M - m) : 0;
        ra[i] = ld4(P.A + (size_t)k * P.lda + m, nv, vecA);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = tid + 256 * i;
      if (!bkm) {
        const int r = s >> 3, c4 = s & 7;
        const int col = n0 + r, k = k0 + c4 * 4;
        const int nv = col < N ? clamp4(kend - k) : 0;
        rb[i] = ld4(P.B + (size_t)col * P.ldb + k, nv, vecB);
      } else {
        const int kk = s >> 5, c4 = s & 31;
        const int k = k0 + kk, n = n0 + c4 * 4;
        const int nv = k < kend ? clamp4(N - n) : 0;
        rb[i] = ld4(P.B + (size_t)k * P.ldb + n, nv, vecB);
      }
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float bsum = 0.f;   // bias-gradient partial (A k-major, first N tile)
  const bool do_bsum = (P.bias_grad != nullptr) && akm && (tn == 0);

  if constexpr (PREC == 0) {
    float* As = reinterpret_cast<float*>(smem_raw);
    float* Bs = As + A_F32;
    auto sstore = [&]() {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int s = tid + 256 * i;
        if (!akm) *reinterpret_cast<float4*>(As + (s >> 3) * LDM + (s & 7) * 4) = ra[i];
        else      *reinterpret_cast<float4*>(As + (s >> 4) * LDKA + (s & 15) * 4) = ra[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = tid + 256 * i;
        if (!bkm) *reinterpret_cast<float4*>(Bs + (s >> 3) * LDM + (s & 7) * 4) = rb[i];
        else      *reinterpret_cast<float4*>(Bs + (s >> 5) * LDKB + (s & 31) * 4) = rb[i];
      }
    };

    gload(kbeg);
    sstore();
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = k0 + BK < kend;
      if (more) gload(k0 + BK);
      float a[16], b[2][16];
      if (!akm) {
        const float* p = As + (wr * 32 + l31) * LDM + 16 * h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(p + 4 * i);
          a[4 * i] = v.x; a[4 * i + 1] = v.y; a[4 * i + 2] = v.z; a[4 * i + 3] = v.w;
        }
      } else {
        const float* p = As + (16 * h) * LDKA + wr * 32 + l31;
#pragma unroll
        for (int s = 0; s < 16; ++s) a[s] = p[s * LDKA];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = wc * 64 + j * 32 + l31;
        if (!bkm) {
          const float* p = Bs + c * LDM + 16 * h;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(p + 4 * i);
            b[j][4 * i] = v.x; b[j][4 * i + 1] = v.y; b[j][4 * i + 2] = v.z; b[j][4 * i + 3] = v.w;
          }
        } else {
          const float* p = Bs + (16 * h) * LDKB + c;
#pragma unroll
          for (int s = 0; s < 16; ++s) b[j][s] = p[s * LDKB];
        }
      }
      if (do_bsum && tid < BM) {
#pragma unroll 8
        for (int kk = 0; kk < BK; ++kk) bsum += As[kk * LDKA + tid];
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[0][s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[1][s], acc[1], 0, 0, 0);
      }
      __syncthreads();
      if (more) { sstore(); __syncthreads(); }
    }
  } else {
    unsigned short* As = reinterpret_cast<unsigned short*>(smem_raw);
    unsigned short* Bs = As + A_BF16;
    // Staging to the bf16 [row][k] image.  m-major sources: a float4 is 4 consecutive k of one
    // row -> one 8-B store.  k-major sources: the thread's float4s are 4 consecutive rows(m) at one
    // k; slots tid+256*i walk k in steps (A: 16, B: 8), so element stores are 2-B scatters.
    auto sstore = [&]() {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int s = tid + 256 * i;
        if (!akm) {
          uint2 pk;
          pk.x = (uint32_t)f2bf(ra[i].x) | ((uint32_t)f2bf(ra[i].y) << 16);
          pk.y = (uint32_t)f2bf(ra[i].z) | ((uint32_t)f2bf(ra[i].w) << 16);
          *reinterpret_cast<uint2*>(As + (s >> 3) * LDH + (s & 7) * 4) = pk;
        } else {
          const int kk = s >> 4, m = (s & 15) * 4;
          As[(m + 0) * LDH + kk] = f2bf(ra[i].x);
          As[(m + 1) * LDH + kk] = f2bf(ra[i].y);
          As[(m + 2) * LDH + kk] = f2bf(ra[i].z);
          As[(m + 3) * LDH + kk] = f2bf(ra[i].w);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = tid + 256 * i;
        if (!bkm) {
          uint2 pk;
          pk.x = (uint32_t)f2bf(rb[i].x) | ((uint32_t)f2bf(rb[i].y) << 16);
          pk.y = (uint32_t)f2bf(rb[i].z) | ((uint32_t)f2bf(rb[i].w) << 16);
          *reinterpret_cast<uint2*>(Bs + (s >> 3) * LDH + (s & 7) * 4) = pk;
        } else {
          const int kk = s >> 5, n = (s & 31) * 4;
          Bs[(n + 0) * LDH + kk] = f2bf(rb[i].x);
          Bs[(n + 1) * LDH + kk] = f2bf(rb[i].y);
          Bs[(n + 2) * LDH + kk] = f2bf(rb[i].z);
          Bs[(n + 3) * LDH + kk] = f2bf(rb[i].w);
        }
      }
    };

    gload(kbeg);
    sstore();
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = k0 + BK < kend;
      if (more) gload(k0 + BK);
      bf16x8 a[2], b[2][2];
      {
        const unsigned short* p = As + (wr * 32 + l31) * LDH + 8 * h;
        a[0] = *reinterpret_cast<const bf16x8*>(p);
        a[1] = *reinterpret_cast<const bf16x8*>(p + 16);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned short* p = Bs + (wc * 64 + j * 32 + l31) * LDH + 8 * h;
        b[j][0] = *reinterpret_cast<const bf16x8*>(p);
        b[j][1] = *reinterpret_cast<const bf16x8*>(p + 16);
      }
      if (do_bsum && tid < BM) {
        // bias gradient from the bf16 image (what the MFMA sees), summed in fp32
        const unsigned short* p = As + tid * LDH;
#pragma unroll 8
        for (int kk = 0; kk < BK; ++kk) bsum += __uint_as_float((uint32_t)p[kk] << 16);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[0][s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[1][s], acc[1], 0, 0, 0);
      }
      __syncthreads();
      if (more) { sstore(); __syncthreads(); }
    }
  }

  if (do_bsum && tid < BM && m0 + tid < M) atomicAdd(P.bias_grad + m0 + tid, bsum);

  // ---- epilogue.  32x32 accumulator: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int flags = P.flags;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc * 64 + j * 32 + l31;
    if (col >= N) continue;
    const float bv = P.bias ? P.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row >= M) continue;
      float v = acc[j][r] + bv;
      if (flags & GF_RELU) v = fmaxf(v, 0.f);
      if ((flags & GF_DROPOUT) && gb.drop.p > 0.f)
        v *= drop_mult(gb.drop, P.drop_site, (uint32_t)row * (uint32_t)N + (uint32_t)col);
      if (flags & GF_RELU_BWD) {
        v *= (P.res[(size_t)row * P.ldr + col] > 0.f) ? P.aux_scale : 0.f;
      } else if (P.res) {
        if (flags & GF_RES_BCAST) {
          int sb; float inv;
          if (P.row_sample) { sb = P.row_sample[row]; inv = P.inv_nr[sb]; }
          else { sb = row / P.uniform_n; inv = 1.0f / (float)P.uniform_n; }
          v += P.res[(size_t)sb * P.ldr + col] * inv;
        } else {
          v += P.res[(size_t)row * P.ldr + col];
        }
      }
      if (flags & GF_SIGMOID) v = 1.0f / (1.0f + __expf(-v));
      float* dst = P.C + (size_t)row * P.ldc + col;
      if (flags & GF_ATOMIC) atomicAdd(dst, v);
      else *dst = v;
    }
  }
}

}  // namespace

namespace {
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> ev;     // 2 per launch
  std::vector<double> flops;
  size_t used = 0;
} g_prof;
}  // namespace

int gemm_prof_begin(int max_launches) {
  if (max_launches < 1) return (int)hipErrorInvalidValue;
  while (g_prof.ev.size() < (size_t)2 * max_launches) {
    hipEvent_t e;
    hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) return (int)r;
    g_prof.ev.push_back(e);
  }
  g_prof.flops.clear();
  g_prof.used = 0;
  g_prof.on = true;
  return 0;
}

int gemm_prof_end(double* total_ms, int* launches, double* total_flops) {
  g_prof.on = false;
  double ms = 0.0, fl = 0.0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    hipError_t r = hipEventSynchronize(g_prof.ev[2 * i + 1]);
    if (r != hipSuccess) return (int)r;
    float t = 0.f;
    r = hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
    if (r != hipSuccess) return (int)r;
    ms += t; fl += g_prof.flops[i];
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = (int)g_prof.used;
  if (total_flops) *total_flops = fl;
  return 0;
}

int launch_gemm_batch(GemmBatch& gb, int precision, hipStream_t stream) {
  if (gb.n <= 0) return 0;
  // tiles without split-K
  int base_tiles = 0;
  for (int i = 0; i < gb.n; ++i) {
    GemmProb& p = gb.p[i];
    p.tiles_n = (p.N + BN - 1) / BN;
    base_tiles += ((p.M + BM - 1) / BM) * p.tiles_n;
  }
  int total = 0;
  for (int i = 0; i < gb.n; ++i) {
    GemmProb& p = gb.p[i];
    const int tiles = ((p.M + BM - 1) / BM) * p.tiles_n;
    int ksplit = 1;
    if ((p.flags & GF_ATOMIC) && (p.flags & GF_A_KMAJOR) && p.K > 4 * BK) {
      // weight-gradient GEMM: small output, long contraction -> split K to fill the chip
      const int ktiles = (p.K + BK - 1) / BK;
      int want = (1024 + base_tiles - 1) / base_tiles;          // aim at ~1024 blocks per launch
      const int max_split = (ktiles + 3) / 4;                      // keep >= 4 K-tiles per block
      ksplit = want < 1 ? 1 : (want > max_split ? max_split : want);
    }
    const int ktiles = (p.K + BK - 1) / BK;
    const int per = (ktiles + ksplit - 1) / ksplit;
    p.kchunk = per * BK;
    p.ksplit = (ktiles + per - 1) / per;
    if (p.ksplit < 1) p.ksplit = 1;
    p.tile_begin = total;
    total += tiles * p.ksplit;
  }
  if (total == 0) return 0;
  const size_t lds = precision == 0 ? (size_t)(A_F32 + B_F32) * 4 : (size_t)(A_BF16 + B_BF16) * 2;
  const bool prof = g_prof.on && 2 * (g_prof.used + 1) <= g_prof.ev.size();
  if (prof) {
    double fl = 0.0;
    for (int i = 0; i < gb.n; ++i) fl += 2.0 * gb.p[i].M * (double)gb.p[i].N * gb.p[i].K;
    g_prof.flops.push_back(fl);
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used], stream);
  }
  if (precision == 0)
    hipLaunchKernelGGL(gemm_grouped_kernel<0>, dim3(total), dim3(256), lds, stream, gb, total);
  else
    hipLaunchKernelGGL(gemm_grouped_kernel<1>, dim3(total), dim3(256), lds, stream, gb, total);
  if (prof) {
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], stream);
    ++g_prof.used;
  }
  return (int)hipGetLastError();
}
