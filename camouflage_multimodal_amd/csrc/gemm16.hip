// Grouped GEMM on bf16-resident operands for gfx950 (see gemm16.h for the contract).
//
// Block tile 64(M) x 128(N), K step 64, 4 waves as 2(M) x 2(N), each wave two 32x32 fp32 accumulators on
// v_mfma_f32_32x32x16_bf16.  Global -> registers -> LDS staging with PIPE register stages (16-byte loads
// from a wave-uniform tile pointer plus a loop-invariant 32-bit lane offset: no address arithmetic, no
// masks and no conversions inside the K loop), double-buffered LDS, one barrier per K step.
//
// LDS images (bank rules: MI355X_MICROARCH.md, LDS):
//   NT  : both operands k-contiguous.  [row][64 k] bf16, dense 128-byte rows whose eight 16-byte chunks are
//         XOR-swizzled by (row >> 1) & 7: the sixteen rows of a ds_read_b128 lane group ({0-3,12-15,20-27}, ...)
//         then cover sixteen distinct 16-byte bank quads, a row's eight chunks still fill one 128-byte line for
//         the stores, and two stages are 48 KB -- three blocks per CU for the forward (NT-only) launches.
//   TN  : both operands are stored [k][m] / [k][n] (the contraction runs over ROWS of dy and x).  The tile is
//         written as it is read from HBM, [k row][m] with pitch 192 B (A, 64 m) / 320 B (B, 128 n), and the
//         MFMA fragments (8 consecutive k per lane) come out of ds_read_b64_tr_b16, the hardware transposing
//         read: a 16-lane group fetches 4 k rows x 16 columns and each lane receives its column's 4 k values.
//         Pitch = 64 mod 256 puts the 4 rows of a 32-lane group on the four 64-byte bank quarters.
// Epilogue: NT accumulates C^T (weight-side operand as MFMA "A"), so a lane owns 4 consecutive columns of
// one output row -> 16-byte fp32 stores and 8-byte bf16 stores; TN keeps C orientation for its fp32 atomics
// (a wave-instruction covers 128 contiguous bytes of two rows).
#include "gemm16.h"


namespace {

constexpr int BM = 64, BN = 128, BK = 64;
constexpr int PM = 128;                 // row pitch (bytes) of a k-contiguous image: dense rows, chunks XOR-swizzled
constexpr int PKA = 192, PKB = 320;     // k-row pitch (bytes) of the k-major images
constexpr int A_NT_BYTES = BM * PM;     // 8192
constexpr int A_TN_BYTES = BK * PKA;    // 12288
constexpr int NT_BUF_BYTES = (BM + BN) * PM;   // 24576: two stages = 48 KB, three blocks per CU
constexpr int BUF_BYTES = 32768;        // TN stage: 12288 + 64*320

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Stage { u32x4 a[2], b[4]; };

// act chunk (8 bf16) + 8 gradient values -> 8 bf16 of (act > 0 ? g * scale : 0)
__device__ __forceinline__ u32x4 virt_chunk(u32x4 act, float4 g0, float4 g1, float scale) {
  auto sel = [&](uint32_t bits, float g) { return __uint_as_float(bits) > 0.f ? g * scale : 0.f; };
  u32x4 o;
  o.x = pack2(sel(act.x << 16, g0.x), sel(act.x & 0xFFFF0000u, g0.y));
  o.y = pack2(sel(act.y << 16, g0.z), sel(act.y & 0xFFFF0000u, g0.w));
  o.z = pack2(sel(act.z << 16, g1.x), sel(act.z & 0xFFFF0000u, g1.y));
  o.w = pack2(sel(act.w << 16, g1.z), sel(act.w & 0xFFFF0000u, g1.w));
  return o;
}

__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// The staged-K loop shared by both layouts.  Straight-line on purpose (see gemm.hip, PIPELINE_LOOP): every
// gload is unconditional (tile index clamped into the block's range), tiles past `nt` are reloads of the
// last tile that get written to LDS and never multiplied.
#define G16_PIPELINE_LOOP                                                                      \
  _Pragma("unroll") for (int j = 0; j < PIPE; ++j) gloadi(j, j);                               \
  sstorei(0, 0);                                                                               \
  gloadi(0, PIPE);                                                                             \
  __syncthreads();                                                                             \
  for (int t = 0; t < nt; t += PIPE) {                                                         \
    _Pragma("unroll") for (int j = 0; j < PIPE; ++j) {                                         \
      sstorei((j + 1) % PIPE, (j + 1) & 1);                                                    \
      gloadi((j + 1) % PIPE, t + j + 1 + PIPE);                                                \
      if (t + j < nt) compute(j & 1);                                                          \
      __syncthreads();                                                                         \
    }                                                                                          \
  }

// ------------------------------------------------------------------------------------------ NT
template <int PIPE>
__device__ __forceinline__ void body_nt(const Gemm16Batch& gb, const Gemm16Prob& P, int m0, int n0, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int M = P.M, N = P.N;
  const int nt = P.K / BK;

  // thread -> 16-byte chunk slots: slot s = tid + 256 i is (row s >> 3, chunk s & 7); 8 neighbouring lanes
  // move one row's 128 contiguous bytes
  uint32_t goa[2], gob[4];
  int la[2], lb[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int s = tid + 256 * i, row = s >> 3, c = s & 7;
    goa[i] = ((uint32_t)min(m0 + row, M - 1) * (uint32_t)P.lda + 8u * c) * 2u;
    la[i] = row * PM + 16 * (c ^ ((row >> 1) & 7));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int s = tid + 256 * i, row = s >> 3, c = s & 7;
    gob[i] = ((uint32_t)min(n0 + row, N - 1) * (uint32_t)P.ldb + 8u * c) * 2u;
    lb[i] = A_NT_BYTES + row * PM + 16 * (c ^ ((row >> 1) & 7));
  }
  const char* Ab = reinterpret_cast<const char*>(P.A);
  const char* Bb = reinterpret_cast<const char*>(P.B);

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  Stage st[PIPE];

  auto gload = [&](Stage& r, int kt) {
    const int k = min(kt, nt - 1);                                   // wave-uniform
    const char* a = Ab + (size_t)k * (BK * 2);
    const char* b = Bb + (size_t)k * (BK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) r.a[i] = *reinterpret_cast<const u32x4*>(a + goa[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.b[i] = *reinterpret_cast<const u32x4*>(b + gob[i]);
  };
  auto sstore = [&](const Stage& r, int buf) {
    char* base = smem + buf * NT_BUF_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(base + la[i]) = r.a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(base + lb[i]) = r.b[i];
  };
  // lane (l31, h) of MFMA step s holds k = 16 s + 8 h .. + 7 of its row: logical chunk 2 s + h, swizzled by the row
  // (rows r and r + 32 share the swizzle, so the second B fragment is a fixed 32 rows further)
  const int ra = wr * 32 + l31, rb = wc * 64 + l31;
  int fa[4], fb[4];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    fa[s_] = ra * PM + 16 * ((2 * s_ + h) ^ ((ra >> 1) & 7));
    fb[s_] = A_NT_BYTES + rb * PM + 16 * ((2 * s_ + h) ^ ((rb >> 1) & 7));
  }
  auto compute = [&](int buf) {
    const char* base = smem + buf * NT_BUF_BYTES;
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(base + fa[s_]);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(base + fb[s_]);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(base + fb[s_] + 32 * PM);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a, acc[0], 0, 0, 0);   // C^T
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a, acc[1], 0, 0, 0);
    }
  };
  auto gloadi = [&](int j, int kt) { gload(st[j], kt); };
  auto sstorei = [&](int j, int buf) { sstore(st[j], buf); };
  G16_PIPELINE_LOOP

  // ---- epilogue.  C^T orientation: lane&31 = output row, registers 4g..4g+3 = columns c0..c0+3,
  // c0 = 32 j + 8 g + 4 (lane>>5)
  const int row = m0 + wr * 32 + l31;
  const bool rok = row < M;
  const int rowc = rok ? row : M - 1;
  const int flags = P.flags;
  const float floor_ = (flags & GF_RELU) ? 0.f : -INFINITY;
  const bool dodrop = (flags & GF_DROPOUT) && gb.drop.p > 0.f;
  const float* rrow = nullptr; float rscale = 1.f;
  if (P.res) {
    if (flags & GF_RES_BCAST) {
      int sb;
      if (P.row_sample) { sb = P.row_sample[rowc]; rscale = P.inv_nr[sb]; }
      else { sb = rowc / P.uniform_n; rscale = 1.0f / (float)P.uniform_n; }
      rrow = P.res + (size_t)sb * P.ldr;
    } else {
      rrow = P.res + (size_t)rowc * P.ldr;
    }
  }
  float* crow = P.C ? P.C + (size_t)rowc * P.ldc : nullptr;
  unsigned short* hrow = P.C16 ? P.C16 + (size_t)rowc * P.ldc16 : nullptr;
  const bool cm = P.colmean != nullptr;                       // block-uniform
  constexpr int TP = 132;                                     // fp32 tile pitch: 16-byte row stores stay conflict-free
  float* tile = reinterpret_cast<float*>(smem);               // 64 x 132 floats (the K loop is done with the LDS)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = n0 + wc * 64 + j * 32 + 8 * g + 4 * h;
      float4 v = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
      if (c0 < N) {
        if (P.bias) { const float4 bv = *reinterpret_cast<const float4*>(P.bias + c0); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
        v.x = fmaxf(v.x, floor_); v.y = fmaxf(v.y, floor_); v.z = fmaxf(v.z, floor_); v.w = fmaxf(v.w, floor_);
        if (dodrop) {
          const uint32_t idx = (uint32_t)rowc * (uint32_t)N + (uint32_t)c0;
          v.x *= drop_mult(gb.drop, P.drop_site, idx); v.y *= drop_mult(gb.drop, P.drop_site, idx + 1);
          v.z *= drop_mult(gb.drop, P.drop_site, idx + 2); v.w *= drop_mult(gb.drop, P.drop_site, idx + 3);
        }
        if (rrow) {
          const float4 rv = *reinterpret_cast<const float4*>(rrow + c0);
          if (flags & GF_RELU_BWD) {
            v.x *= rv.x > 0.f ? P.aux_scale : 0.f; v.y *= rv.y > 0.f ? P.aux_scale : 0.f;
            v.z *= rv.z > 0.f ? P.aux_scale : 0.f; v.w *= rv.w > 0.f ? P.aux_scale : 0.f;
          } else {
            v.x = fmaf(rv.x, rscale, v.x); v.y = fmaf(rv.y, rscale, v.y); v.z = fmaf(rv.z, rscale, v.z); v.w = fmaf(rv.w, rscale, v.w);
          }
        }
        if (rok) {
          if (crow) *reinterpret_cast<float4*>(crow + c0) = v;
          if (hrow) *reinterpret_cast<uint2*>(hrow + c0) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
        }
      }
      if (cm) *reinterpret_cast<float4*>(tile + (wr * 32 + l31) * TP + (c0 - n0)) = v;
    }
  if (cm) {
    // per-sample column means of the tile: the rows' sample ids go to LDS once; thread (column t, row half) walks 32
    // rows of the fp32 tile, flushing whenever the sample changes (a 64-row tile rarely meets more than one
    // boundary); the two halves meet in LDS when the whole tile is one sample -> one atomic per column per tile
    int* srow = reinterpret_cast<int*>(tile + 64 * TP);
    float* hsum = tile + 64 * TP + 64;
    if (tid < 64) { const int rr = m0 + tid; srow[tid] = rr < M ? (P.row_sample ? P.row_sample[rr] : rr / P.uniform_n) : -1; }
    __syncthreads();
    const int t = tid & 127, half = tid >> 7, col = n0 + t;
    const int s_first = srow[0];
    const bool one = srow[63] == s_first || (srow[63] < 0 && srow[min(M - 1 - m0, 63)] == s_first);   // block-uniform
    float sum = 0.f; int cur = -1;
    for (int r = 0; r < 32; ++r) {
      const int sm = srow[half * 32 + r];
      if (sm < 0) break;
      if (sm != cur) {
        if (cur >= 0 && col < N) atomicAdd(P.colmean + (size_t)cur * P.ldm + col, sum * (P.row_sample ? P.inv_nr[cur] : 1.0f / (float)P.uniform_n));
        sum = 0.f; cur = sm;
      }
      sum += tile[(half * 32 + r) * TP + t];
    }
    if (one) {
      if (half == 1) hsum[t] = sum;
      __syncthreads();
      if (half == 0 && col < N && cur >= 0)
        atomicAdd(P.colmean + (size_t)cur * P.ldm + col, (sum + hsum[t]) * (P.row_sample ? P.inv_nr[cur] : 1.0f / (float)P.uniform_n));
    } else if (cur >= 0 && col < N) {
      atomicAdd(P.colmean + (size_t)cur * P.ldm + col, sum * (P.row_sample ? P.inv_nr[cur] : 1.0f / (float)P.uniform_n));
    }
  }
}

// ------------------------------------------------------------------------------------------ TN
template <int PIPE, bool VIRT>
__device__ __forceinline__ void body_tn(const Gemm16Prob& P, int m0, int n0, int kt0, int nt, char* smem, int exp = 0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int M = P.M, N = P.N;

  // slots: A tile = 64 k rows x 8 chunks (8 m each), B tile = 64 k rows x 16 chunks (8 n each)
  uint32_t goa[2], gob[4];
  int la[2], lb[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int s = tid + 256 * i, kr = s >> 3, c = s & 7;
    goa[i] = ((uint32_t)kr * (uint32_t)P.lda + (uint32_t)min(m0 + 8 * c, M - 8)) * 2u;
    la[i] = kr * PKA + 16 * c;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int s = tid + 256 * i, kr = s >> 4, c = s & 15;
    gob[i] = ((uint32_t)kr * (uint32_t)P.ldb + (uint32_t)min(n0 + 8 * c, N - 8)) * 2u;
    lb[i] = A_TN_BYTES + kr * PKB + 16 * c;
  }
  const char* Ab = reinterpret_cast<const char*>(P.A);
  const char* Bb = reinterpret_cast<const char*>(P.B);
  const size_t astep = (size_t)BK * P.lda * 2, bstep = (size_t)BK * P.ldb * 2;

  f32x16 acc[2], accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; accb[r] = 0.f; }
  const bool do_bsum = P.bias_grad != nullptr && n0 == 0 && wc == 0;     // wave-uniform
  const short one = (short)0x3F80;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};
  Stage st[PIPE];
  int sa[PIPE][2];                        // VIRT: sample of each staged A row
  // VIRT: the tile's 64 columns of the scaled per-sample gradient rows live in LDS behind the two stage buffers
  float* gs = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);
  if constexpr (VIRT) {
    const int ns = P.uniform_n > 0 && !P.row_sample ? (P.K + P.uniform_n - 1) / P.uniform_n : P.ldc16;   // ldc16 carries the sample count
    for (int i = tid; i < ns * 64; i += 256) {
      const int sm = i >> 6, f = i & 63, c = min(m0 + f, M - 1);
      const float inv = P.row_sample ? P.inv_nr[sm] : 1.0f / (float)P.uniform_n;
      gs[i] = P.virt_g[(size_t)sm * P.ldg + c] * inv * P.aux_scale;
    }
    __syncthreads();
  }

  auto gload = [&](Stage& r, int (&srow)[2], int kt) {
    const int k = kt0 + min(kt, nt - 1);                             // wave-uniform
    const char* a = Ab + (size_t)k * astep;
    const char* b = Bb + (size_t)k * bstep;
#pragma unroll
    for (int i = 0; i < 2; ++i) r.a[i] = *reinterpret_cast<const u32x4*>(a + goa[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.b[i] = *reinterpret_cast<const u32x4*>(b + gob[i]);
    if constexpr (VIRT) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int node = min(k * BK + ((tid + 256 * i) >> 3), P.K - 1);
        srow[i] = P.row_sample ? P.row_sample[node] : node / P.uniform_n;
      }
    }
  };
  auto sstore = [&](const Stage& r, const int (&srow)[2], int buf) {
    char* base = smem + buf * BUF_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 a = r.a[i];
      if constexpr (VIRT) {
        const float* g = gs + srow[i] * 64 + 8 * ((tid + 256 * i) & 7);
        a = virt_chunk(a, *reinterpret_cast<const float4*>(g), *reinterpret_cast<const float4*>(g + 4), 1.0f);
      }
      *reinterpret_cast<u32x4*>(base + la[i]) = a;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(base + lb[i]) = r.b[i];
  };
  // transposing reads: lane = 16 g + 4 q + p of its 32-lane half supplies (k row .. + q, columns 16 g + 4 p ..)
  // and receives column 16 g + (lane & 15) = lane & 31 of those 4 k rows
  const int g = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  const int fa = (8 * h + q) * PKA + 64 * wr + 32 * g + 8 * p;
  const int fb = A_TN_BYTES + (8 * h + q) * PKB + 128 * wc + 32 * g + 8 * p;
  auto compute = [&](int buf) {
    const char* base = smem + buf * BUF_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const char* pa = base + fa + 16 * s * PKA;
      const char* pb = base + fb + 16 * s * PKB;
      const s16x4 a0 = lds_tr16(pa), a1 = lds_tr16(pa + 4 * PKA);
      const s16x4 b00 = lds_tr16(pb), b01 = lds_tr16(pb + 4 * PKB);
      const s16x4 b10 = lds_tr16(pb + 64), b11 = lds_tr16(pb + 64 + 4 * PKB);
      const bf16x8 a = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const bf16x8 b0 = {b00.x, b00.y, b00.z, b00.w, b01.x, b01.y, b01.z, b01.w};
      const bf16x8 b1 = {b10.x, b10.y, b10.z, b10.w, b11.x, b11.y, b11.z, b11.w};
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[1], 0, 0, 0);
      if (do_bsum) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, accb, 0, 0, 0);   // row sums of dy^T
    }
  };
  auto gloadi = [&](int j, int kt) { gload(st[j], sa[j], kt); };
  auto sstorei = [&](int j, int buf) { sstore(st[j], sa[j], buf); };
  G16_PIPELINE_LOOP

  // C orientation: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int rowb = m0 + wr * 32 + 4 * h;
  if ((exp & 1) && acc[0][0] != 12345.f) return;            // (experiment: no epilogue)
  if (do_bsum && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rowb + (r & 3) + 8 * (r >> 2);
      if (row < M) { atomicAdd(P.bias_grad + row, accb[r]); if (P.bias_grad2) atomicAdd(P.bias_grad2 + row, accb[r]); }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc * 64 + j * 32 + l31;
    if (col >= N) continue;
    float* cp = P.C + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rowb + (r & 3) + 8 * (r >> 2);
      if (row < M) atomicAdd(cp + (size_t)row * P.ldc, acc[j][r]);
    }
  }
}


// ------------------------------------------------------------------------------------------ NT, whole-row tiles
// 32 rows x 256 columns per block (4 waves side by side, 64 columns each), so a row's LayerNorm statistics are
// inside the block: the LayerNorm forward rides on the out-projection, the LayerNorm backward on the FFN input
// gradient, and neither needs a launch or a round trip of its input through HBM.
constexpr int RM = 32, RN = 256;
constexpr int A_ROW_BYTES = RM * PM;                 // 4096
constexpr int ROW_BUF_BYTES = (RM + RN) * PM;        // 36864 per stage
constexpr int TPR = RN + 4;                          // fp32 tile pitch of the column passes
struct StageR { u32x4 a; u32x4 b[8]; float4 g0, g1; };     // g0, g1: VIRT only (the 8 gradient values of the A chunk)

template <int PIPE, bool VIRT>
__device__ __forceinline__ void body_nt_row(const Gemm16Batch& gb, const Gemm16Prob& P, int m0, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int M = P.M, N = RN;
  const int nt = P.K / BK;
  uint32_t goa, gob[8];
  int la, lb[8];
  {
    const int row = tid >> 3, c = tid & 7;
    goa = ((uint32_t)min(m0 + row, M - 1) * (uint32_t)P.lda + 8u * c) * 2u;
    la = row * PM + 16 * (c ^ ((row >> 1) & 7));
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int s_ = tid + 256 * i, row = s_ >> 3, c = s_ & 7;
    gob[i] = ((uint32_t)row * (uint32_t)P.ldb + 8u * c) * 2u;
    lb[i] = A_ROW_BYTES + row * PM + 16 * (c ^ ((row >> 1) & 7));
  }
  const char* Ab = reinterpret_cast<const char*>(P.A);
  const char* Bb = reinterpret_cast<const char*>(P.B);
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  StageR st[PIPE];
  // VIRT: this thread's A row is node min(m0 + tid/8, M-1): its sample's gradient row and scale are loop-invariant
  const float* vg = nullptr; float vscale = 0.f;
  if constexpr (VIRT) {
    const int node = min(m0 + (tid >> 3), M - 1);
    const int sm = P.row_sample ? P.row_sample[node] : node / P.uniform_n;
    vg = P.virt_g + (size_t)sm * P.ldg + 8 * (tid & 7);
    vscale = (P.row_sample ? P.inv_nr[sm] : 1.0f / (float)P.uniform_n) * P.aux_scale;
  }
  auto gload = [&](StageR& r, int kt) {
    const int k = min(kt, nt - 1);
    const char* a = Ab + (size_t)k * (BK * 2);
    const char* b = Bb + (size_t)k * (BK * 2);
    r.a = *reinterpret_cast<const u32x4*>(a + goa);
    if constexpr (VIRT) { r.g0 = *reinterpret_cast<const float4*>(vg + k * BK); r.g1 = *reinterpret_cast<const float4*>(vg + k * BK + 4); }
#pragma unroll
    for (int i = 0; i < 8; ++i) r.b[i] = *reinterpret_cast<const u32x4*>(b + gob[i]);
  };
  auto sstore = [&](const StageR& r, int buf) {
    char* base = smem + buf * ROW_BUF_BYTES;
    if constexpr (VIRT) *reinterpret_cast<u32x4*>(base + la) = virt_chunk(r.a, r.g0, r.g1, vscale);
    else *reinterpret_cast<u32x4*>(base + la) = r.a;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(base + lb[i]) = r.b[i];
  };
  const int rb = w * 64 + l31;
  int fa[4], fb[4];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    fa[s_] = l31 * PM + 16 * ((2 * s_ + h) ^ ((l31 >> 1) & 7));
    fb[s_] = A_ROW_BYTES + rb * PM + 16 * ((2 * s_ + h) ^ ((rb >> 1) & 7));
  }
  auto compute = [&](int buf) {
    const char* base = smem + buf * ROW_BUF_BYTES;
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(base + fa[s_]);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(base + fb[s_]);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(base + fb[s_] + 32 * PM);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a, acc[0], 0, 0, 0);   // C^T: lane = row
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a, acc[1], 0, 0, 0);
    }
  };
  auto gloadi = [&](int j, int kt) { gload(st[j], kt); };
  auto sstorei = [&](int j, int buf) { sstore(st[j], buf); };
  G16_PIPELINE_LOOP

  // ---- epilogue: lane (l31, h) holds row m0 + l31, columns 64 w + 32 j + 8 g + 4 h + i (register 4 g + i of acc[j])
  const int row = m0 + l31;
  const bool rok = row < M;
  const int rowc = rok ? row : M - 1;
  float* red = reinterpret_cast<float*>(smem);        // [2][4][32]
  float* tile0 = red + 256;                           // [32][TPR]
  float* tile1 = tile0 + RM * TPR;                    // [32][TPR] (backward only)
  const float invN = 1.0f / (float)N;
  // sum of the lane's 32 values + the other half-wave's = this wave's 64 columns of the row; then the 4 waves
  auto row_total = [&](float part, int slot) {
    part += __shfl_xor(part, 32, 64);
    if (h == 0) red[slot * 128 + w * 32 + l31] = part;
    __syncthreads();
    return (red[slot * 128 + l31] + red[slot * 128 + 32 + l31]) + (red[slot * 128 + 64 + l31] + red[slot * 128 + 96 + l31]);
  };
  int sb = 0; float rscale = 1.f;
  if (P.row_sample) { sb = P.row_sample[rowc]; rscale = P.inv_nr[sb]; }
  else if (P.uniform_n > 0) { sb = rowc / P.uniform_n; rscale = 1.0f / (float)P.uniform_n; }
  if (P.ln_mode == 1) {
    const float* rrow = P.res ? P.res + (size_t)rowc * P.ldr : nullptr;
    float4 u[2][4];
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = w * 64 + j * 32 + 8 * g + 4 * h;
        float4 v = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
        if (P.bias) { const float4 bv = *reinterpret_cast<const float4*>(P.bias + c0); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
        if (rrow) { const float4 rv = *reinterpret_cast<const float4*>(rrow + c0); v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
        u[j][g] = v;
        part += (v.x + v.y) + (v.z + v.w);
        if (rok && P.C) *reinterpret_cast<float4*>(P.C + (size_t)row * P.ldc + c0) = v;
      }
    const float mean = row_total(part, 0) * invN;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4& v = u[j][g];
        v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
        sq = fmaf(v.x, v.x, sq); sq = fmaf(v.y, v.y, sq); sq = fmaf(v.z, v.z, sq); sq = fmaf(v.w, v.w, sq);
      }
    const float rstd = 1.0f / sqrtf(row_total(sq, 1) * invN + 1e-5f);
    if (rok && w == 0 && h == 0 && P.ln_stats) { P.ln_stats[2 * row] = mean; P.ln_stats[2 * row + 1] = rstd; }
    __syncthreads();                                   // `red` is read; the tile below overlaps nothing of it
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = w * 64 + j * 32 + 8 * g + 4 * h;
        const float4 gm = *reinterpret_cast<const float4*>(P.ln_gamma + c0), bt = *reinterpret_cast<const float4*>(P.ln_beta + c0);
        float4 y;
        y.x = u[j][g].x * rstd * gm.x + bt.x; y.y = u[j][g].y * rstd * gm.y + bt.y;
        y.z = u[j][g].z * rstd * gm.z + bt.z; y.w = u[j][g].w * rstd * gm.w + bt.w;
        if (rok && P.C16) *reinterpret_cast<uint2*>(P.C16 + (size_t)row * P.ldc16 + c0) = make_uint2(pack2(y.x, y.y), pack2(y.z, y.w));
        if (P.colmean) *reinterpret_cast<float4*>(tile0 + l31 * TPR + c0) = y;
      }
    if (P.colmean) {                                   // per-sample mean pool of y: thread t owns column t
      int* srow = reinterpret_cast<int*>(tile1);
      if (tid < RM) { const int rr = m0 + tid; srow[tid] = rr < M ? (P.row_sample ? P.row_sample[rr] : rr / P.uniform_n) : -1; }
      __syncthreads();
      float sum = 0.f; int cur = -1;
      for (int r = 0; r < RM; ++r) {
        const int sm = srow[r];
        if (sm < 0) break;
        if (sm != cur) {
          if (cur >= 0) atomicAdd(P.colmean + (size_t)cur * P.ldm + tid, sum * (P.row_sample ? P.inv_nr[cur] : 1.0f / (float)P.uniform_n));
          sum = 0.f; cur = sm;
        }
        sum += tile0[r * TPR + tid];
      }
      if (cur >= 0) atomicAdd(P.colmean + (size_t)cur * P.ldm + tid, sum * (P.row_sample ? P.inv_nr[cur] : 1.0f / (float)P.uniform_n));
    }
  } else {
    // LayerNorm backward: g = dy * gamma, du = (g - mean(g) - xhat * mean(g * xhat)) * rstd
    const bool bc = P.flags & GF_RES_BCAST;                                    // per-sample broadcast residual, or per-row
    const float* rrow = P.res ? P.res + (size_t)(bc ? sb : rowc) * P.ldr : nullptr;
    if (!bc) rscale = 1.f;
    const float* xrow = P.ln_x + (size_t)rowc * N;
    const float mean = P.ln_stats[2 * rowc], rstd = P.ln_stats[2 * rowc + 1];
    float4 gq[2][4], xq[2][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = w * 64 + j * 32 + 8 * g + 4 * h;
        float4 dy = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
        if (rrow) { const float4 rv = *reinterpret_cast<const float4*>(rrow + c0); dy.x = fmaf(rv.x, rscale, dy.x); dy.y = fmaf(rv.y, rscale, dy.y); dy.z = fmaf(rv.z, rscale, dy.z); dy.w = fmaf(rv.w, rscale, dy.w); }
        const float4 xv = *reinterpret_cast<const float4*>(xrow + c0);
        const float4 gm = *reinterpret_cast<const float4*>(P.ln_gamma + c0);
        float4 xh, gg;
        xh.x = (xv.x - mean) * rstd; xh.y = (xv.y - mean) * rstd; xh.z = (xv.z - mean) * rstd; xh.w = (xv.w - mean) * rstd;
        gg.x = dy.x * gm.x; gg.y = dy.y * gm.y; gg.z = dy.z * gm.z; gg.w = dy.w * gm.w;
        s1 += (gg.x + gg.y) + (gg.z + gg.w);
        s2 = fmaf(gg.x, xh.x, s2); s2 = fmaf(gg.y, xh.y, s2); s2 = fmaf(gg.z, xh.z, s2); s2 = fmaf(gg.w, xh.w, s2);
        gq[j][g] = gg; xq[j][g] = xh;
        // column-sum operands of dgamma / dbeta (rows past M are excluded by the column pass)
        *reinterpret_cast<float4*>(tile0 + l31 * TPR + c0) = make_float4(dy.x * xh.x, dy.y * xh.y, dy.z * xh.z, dy.w * xh.w);
        *reinterpret_cast<float4*>(tile1 + l31 * TPR + c0) = dy;
      }
    const float m1 = row_total(s1, 0) * invN;
    const float m2 = row_total(s2, 1) * invN;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = w * 64 + j * 32 + 8 * g + 4 * h;
        float4 du;
        du.x = (gq[j][g].x - m1 - xq[j][g].x * m2) * rstd; du.y = (gq[j][g].y - m1 - xq[j][g].y * m2) * rstd;
        du.z = (gq[j][g].z - m1 - xq[j][g].z * m2) * rstd; du.w = (gq[j][g].w - m1 - xq[j][g].w * m2) * rstd;
        if (rok) {
          if (P.C) *reinterpret_cast<float4*>(P.C + (size_t)row * P.ldc + c0) = du;
          if (P.C16) *reinterpret_cast<uint2*>(P.C16 + (size_t)row * P.ldc16 + c0) = make_uint2(pack2(du.x, du.y), pack2(du.z, du.w));
        }
      }
    // (row_total's barriers also ordered the tile stores above before the column pass)
    const int nrow = min(RM, M - m0);
    float a = 0.f, b = 0.f;
    for (int r = 0; r < nrow; ++r) { a += tile0[r * TPR + tid]; b += tile1[r * TPR + tid]; }
    atomicAdd(P.ln_dgamma + tid, a);
    atomicAdd(P.ln_dbeta + tid, b);
  }
}

// ROWS: the launch contains whole-row (fused LayerNorm) problems -> 72 KB of LDS, two blocks per CU; otherwise
// three blocks per CU (forward launches) or two (launches with weight-gradient problems, 64 KB).
template <int PIPE, bool ROWS>
__global__ __launch_bounds__(256, ROWS ? 2 : 3) void gemm16_kernel(const Gemm16Batch gb, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // XCD-aware block remap (bijective): blocks that share an XCD (bid % 8) take a contiguous run of tiles
  int bid = blockIdx.x;
  {
    const int nwg = total_tiles, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    const int first = xcd * q + min(xcd, r);                  // the XCD's run in the plain remap
    if (gb.n_heavy > 0) {
      // the weight-gradient launch: long split-K blocks (RG-side problems) in front, short ones (the 13-row KG side) behind.  One
      // contiguous run per XCD gave XCD 7 all the short blocks and 3 long ones (idle after 16 of the launch's 26 us) and the
      // others 59 long ones each; here every XCD takes an equal share of each kind, still contiguous within the kind
      const int nh = gb.n_heavy, qh = nh >> 3, rh = nh & 7;
      const int hc = qh + (xcd < rh ? 1 : 0), hfirst = xcd * qh + min(xcd, rh);
      bid = j < hc ? hfirst + j : nh + (first - hfirst) + (j - hc);
    } else {
      bid = first + j;
    }
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM16_MAXP; ++i)
    if (i < gb.n && bid >= gb.p[i].tile_begin) pi = i;
  const Gemm16Prob& P = gb.p[pi];
  int t = bid - P.tile_begin;
  if constexpr (ROWS) {
    if (P.ln_mode) {
      if (P.flags & GF_A_VIRT) body_nt_row<PIPE, true>(gb, P, t * RM, smem_raw);
      else                     body_nt_row<PIPE, false>(gb, P, t * RM, smem_raw);
      return;
    }
  }
  // Split-K problems number their blocks K-slice-major: the XCD remap above hands an XCD a contiguous run of block
  // ids, i.e. (with ~8 slices) all output tiles of ONE K slice -- the tiles that re-read the same rows of dy and x
  // then share them through that XCD's L2 instead of fetching them 2-8 times from HBM / Infinity Cache.
  const int tiles = ((P.M + BM - 1) / BM) * P.tiles_n;
  const int ks = t / tiles; t -= ks * tiles;
  const int tn = t % P.tiles_n, tm = t / P.tiles_n;
  if (P.flags & GF_A_KMAJOR) {
    const int ktiles = (P.K + 127) / 128 * 2;                 // K rounded up to 128 rows, in 64-row tiles
    const int per = P.kchunk / BK, kt0 = ks * per;
    if constexpr (ROWS) {
      if (P.flags & GF_A_VIRT) { body_tn<PIPE, true>(P, tm * BM, tn * BN, kt0, min(per, ktiles - kt0), smem_raw); return; }
    }
    body_tn<PIPE, false>(P, tm * BM, tn * BN, kt0, (gb.exp & 2) ? 2 : min(per, ktiles - kt0), smem_raw, gb.exp);
  } else {
    body_nt<PIPE>(gb, P, tm * BM, tn * BN, smem_raw);
  }
}

// ------------------------------------------------------------------------------------------ TN, 128 x 256 tiles on 8 waves
// Weight gradients of long contractions (K = packed rows >= 40 000).  The 64 x 128 tile's K loop waits for operand bytes, not
// for MFMA (0.75 us per 64-row step with two blocks per CU: 24 KB per 1.05 MFLOP); this tile moves 48 KB per 4.2 MFLOP --
// half the bytes per FLOP at the same bytes in flight per CU (one block of 8 waves, two register stages of 2 + 4 chunks per
// thread).  Waves 2 (M) x 4 (N), 64 x 64 each: two A and two B fragments feed four MFMAs per 16-deep step.  The price is
// fewer output tiles, hence more K splits (more fp32 atomics) to fill the chip: taken only where the K loop dominates.
constexpr int GM = 128, GN = 256;
constexpr int GPA = 320, GPB = 576;                 // k-row pitches (bytes): 64 mod 256, like PKA / PKB
constexpr int GA_BYTES = BK * GPA;                  // 20480
constexpr int GBUF_BYTES = GA_BYTES + BK * GPB;     // 57344 per stage

template <int PIPE>
__global__ __launch_bounds__(512) void gemm16_tnbig_kernel(const Gemm16Batch gb, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bid = blockIdx.x;
  {   // XCD-aware remap (see gemm16_kernel): an XCD takes a contiguous run of blocks = the output tiles of a few K slices
    const int nwg = total_tiles, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    bid = xcd * q + min(xcd, r) + j;
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM16_MAXP; ++i)
    if (i < gb.n && bid >= gb.p[i].tile_begin) pi = i;
  const Gemm16Prob& P = gb.p[pi];
  int t = bid - P.tile_begin;
  const int tiles = ((P.M + GM - 1) / GM) * P.tiles_n;
  const int ks = t / tiles; t -= ks * tiles;
  const int n0 = (t % P.tiles_n) * GN, m0 = (t / P.tiles_n) * GM;
  const int ktiles = (P.K + 127) / 128 * 2;                   // K rounded up to 128 rows (pad rows are zero), in 64-row tiles
  const int per = P.kchunk / BK, kt0 = ks * per, nt = min(per, ktiles - kt0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3, l31 = lane & 31, h = lane >> 5;
  const int M = P.M, N = P.N;
  uint32_t goa[2], gob[4];
  int la[2], lb[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {                               // A tile = 64 k rows x 16 chunks (8 m each)
    const int s = tid + 512 * i, kr = s >> 4, c = s & 15;
    goa[i] = ((uint32_t)kr * (uint32_t)P.lda + (uint32_t)min(m0 + 8 * c, M - 8)) * 2u;
    la[i] = kr * GPA + 16 * c;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {                               // B tile = 64 k rows x 32 chunks (8 n each)
    const int s = tid + 512 * i, kr = s >> 5, c = s & 31;
    gob[i] = ((uint32_t)kr * (uint32_t)P.ldb + (uint32_t)min(n0 + 8 * c, N - 8)) * 2u;
    lb[i] = GA_BYTES + kr * GPB + 16 * c;
  }
  const char* Ab = reinterpret_cast<const char*>(P.A);
  const char* Bb = reinterpret_cast<const char*>(P.B);
  const size_t astep = (size_t)BK * P.lda * 2, bstep = (size_t)BK * P.ldb * 2;

  f32x16 acc[2][2], accb[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f; accb[0][r] = 0.f; accb[1][r] = 0.f; }
  const bool do_bsum = P.bias_grad != nullptr && n0 == 0 && wc == 0;     // wave-uniform
  const short one = (short)0x3F80;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};
  Stage st[PIPE];

  auto gloadi = [&](int j, int kt) {
    const int k = kt0 + min(kt, nt - 1);                      // (block-uniform; tiles past nt are reloads that are never multiplied)
    const char* a = Ab + (size_t)k * astep;
    const char* b = Bb + (size_t)k * bstep;
#pragma unroll
    for (int i = 0; i < 2; ++i) st[j].a[i] = *reinterpret_cast<const u32x4*>(a + goa[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) st[j].b[i] = *reinterpret_cast<const u32x4*>(b + gob[i]);
  };
  auto sstorei = [&](int j, int buf) {
    char* base = smem + buf * GBUF_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(base + la[i]) = st[j].a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(base + lb[i]) = st[j].b[i];
  };
  // transposing reads (see body_tn): lane = 16 g + 4 q + p of its half supplies (k row 8 h + q, columns 16 g + 4 p ..)
  const int g = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  const int fa = (8 * h + q) * GPA + 128 * wr + 32 * g + 8 * p;
  const int fb = GA_BYTES + (8 * h + q) * GPB + 128 * wc + 32 * g + 8 * p;
  auto frag = [&](const char* q0, int pitch) {
    const s16x4 lo = lds_tr16(q0), hi = lds_tr16(q0 + 4 * pitch);
    return bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  };
  auto compute = [&](int buf) {
    const char* base = smem + buf * GBUF_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const char* pa = base + fa + 16 * s * GPA;
      const char* pb = base + fb + 16 * s * GPB;
      const bf16x8 a0 = frag(pa, GPA), a1 = frag(pa + 64, GPA);
      const bf16x8 b0 = frag(pb, GPB), b1 = frag(pb + 64, GPB);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
      if (do_bsum) {                                          // row sums of dy^T
        accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, ones, accb[0], 0, 0, 0);
        accb[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, ones, accb[1], 0, 0, 0);
      }
    }
  };
  G16_PIPELINE_LOOP

  // C orientation: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rowb = m0 + 64 * wr + 32 * i + 4 * h;
    if (do_bsum && l31 == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rowb + (r & 3) + 8 * (r >> 2);
        if (row < M) { atomicAdd(P.bias_grad + row, accb[i][r]); if (P.bias_grad2) atomicAdd(P.bias_grad2 + row, accb[i][r]); }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 64 * wc + 32 * j + l31;
      if (col >= N) continue;
      float* cp = P.C + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rowb + (r & 3) + 8 * (r >> 2);
        if (row < M) atomicAdd(cp + (size_t)row * P.ldc, acc[i][j][r]);
      }
    }
  }
}

constexpr int KCAP_MIN = 8;          // shortest split-K chunk (64-row tiles) a sparse launch falls back to
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

thread_local int g_gemm16_exp = 0;
thread_local int g_gemm16_tn_big = -1;             // developer A/B ("tn_big"): -1 by size, 0 never, 1 whenever the launch is weight gradients only
static int launch_tnbig(Gemm16Batch& gb, hipStream_t stream) {
  int total = 0;
  static const int kcaps[] = {512, 384, 256, 192, 128, 96, 64, 48, 32, 24, 16, 12, 8, 6, 4};
  for (int ci = 0;; ++ci) {
    const int kcap = g_gemm16_tn_kcap > 0 ? g_gemm16_tn_kcap : kcaps[ci];
    total = 0;
    for (int i = 0; i < gb.n; ++i) {
      Gemm16Prob& p = gb.p[i];
      if (p.M < 8 || p.N < 8 || p.K < 1 || !p.C || p.C16 || p.bias || p.res || !al16(p.A) || !al16(p.B) || (p.lda & 7) || (p.ldb & 7))
        return (int)hipErrorInvalidValue;
      if ((double)(p.K + 128) * p.lda * 2.0 >= 4.0e9 || (double)(p.K + 128) * p.ldb * 2.0 >= 4.0e9) return (int)hipErrorInvalidValue;
      p.tiles_n = (p.N + GN - 1) / GN;
      const int tiles = ((p.M + GM - 1) / GM) * p.tiles_n;
      const int ktiles = (p.K + 127) / 128 * 2;
      const int per = ktiles < kcap ? ktiles : kcap;                       // even
      p.kchunk = per * BK;
      p.ksplit = (ktiles + per - 1) / per;
      p.tile_begin = total;
      total += tiles * p.ksplit;
    }
    if (g_gemm16_tn_kcap > 0 || total >= 224 || kcap <= 4) break;         // ~ one block per CU
  }
  gb.n_heavy = 0; gb.exp = g_gemm16_exp;
  static const bool attr_ok = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_tnbig_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GBUF_BYTES);
    return true;
  }();
  (void)attr_ok;
  double fl = 0.0;
  for (int i = 0; i < gb.n; ++i) fl += 2.0 * gb.p[i].M * (double)gb.p[i].N * gb.p[i].K;
  const int prof = gemm_prof_open(stream, fl);
  // (four register stages measured the same as two: the K loop is bound by its LDS transposing reads -- SQ_ACTIVE_INST_LDS 80 % of
  // the kernel's busy cycles, no bank conflicts -- not by load latency)
  hipLaunchKernelGGL((gemm16_tnbig_kernel<2>), dim3(total), dim3(512), 2 * GBUF_BYTES, stream, gb, total);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}
thread_local int g_gemm16_balance = 1;             // developer A/B ("tn_balance")
thread_local int g_gemm16_tn_kcap = 0;             // developer A/B (camo_debug_set_option "tn_kcap"): > 0 pins the split-K depth
int launch_gemm16_batch(Gemm16Batch& gb, hipStream_t stream) {
  if (gb.n <= 0) return 0;
  if (gb.n > GEMM16_MAXP) return (int)hipErrorInvalidValue;
  // Split-K depth of the weight-gradient problems, in 64-row tiles per block: the longest of a short ladder that still gives
  // ~2 blocks per CU.  Every block ends in 32 KB of fp32 atomics, which -- not the K loop -- set the launch's length once there
  // are more blocks than that (measured, bf16 training step: B = 16 -> 16 tiles [8: +12 us, 32: +8 us]; B = 64 -> 64 tiles
  // [16: +18 us, 8: +56 us]); a launch that would leave most CUs idle even at 16 (the general schedule's lone dW_rg / dW_kg
  // launch) goes down to 8 to spread over the chip.
  // long weight-gradient contractions: 128 x 256 tiles on 8 waves (gemm16_tnbig_kernel)
  {
    bool all_tn = true; int kmax = 0;
    for (int i = 0; i < gb.n; ++i) {
      const Gemm16Prob& p = gb.p[i];
      all_tn = all_tn && (p.flags & GF_A_KMAJOR) && (p.flags & GF_B_KMAJOR) && !(p.flags & GF_A_VIRT) && !p.ln_mode && (p.M % 8) == 0 && (p.N % 8) == 0;
      kmax = p.K > kmax ? p.K : kmax;
    }
    const bool big = g_gemm16_tn_big < 0 ? kmax >= 40000 : g_gemm16_tn_big > 0;    // (measured: B = 64 [27 k rows] +8 %, B = 128 [55 k] -7 %, B = 256 [125 k] -15 %)
    if (all_tn && big) return launch_tnbig(gb, stream);
  }
  int total = 0, kcap_used = 0;
  static const int kcaps[] = {512, 384, 256, 192, 128, 96, 64, 48, 32, 24, 16, 8};
  for (int ci = 0;; ++ci) {
    const int kcap = g_gemm16_tn_kcap > 0 ? g_gemm16_tn_kcap : kcaps[ci];
    total = 0; kcap_used = kcap;
    for (int i = 0; i < gb.n; ++i) {
      Gemm16Prob& p = gb.p[i];
      const bool akm = p.flags & GF_A_KMAJOR, bkm = p.flags & GF_B_KMAJOR;
      if (akm != bkm || p.M < 1 || p.N < 1 || p.K < 1) return (int)hipErrorInvalidValue;
      if (!al16(p.A) || !al16(p.B) || (p.lda & 7) || (p.ldb & 7)) return (int)hipErrorInvalidValue;
      if ((double)(akm ? p.K + 128 : p.M) * p.lda * 2.0 >= 4.0e9 || (double)(akm ? p.K + 128 : p.N) * p.ldb * 2.0 >= 4.0e9)
        return (int)hipErrorInvalidValue;                                    // 32-bit lane offsets
      p.tiles_n = (p.N + BN - 1) / BN;
      int tiles = ((p.M + BM - 1) / BM) * p.tiles_n;
      if (p.ln_mode) {
        if (akm || p.N != RN || !p.ln_gamma || !al16(p.ln_gamma) || (p.ln_mode == 1 && (!p.ln_beta || !al16(p.ln_beta))) ||
            (p.ln_mode == 2 && (!p.ln_x || !al16(p.ln_x) || !p.ln_stats || !p.ln_dgamma || !p.ln_dbeta)) ||
            (p.flags & (GF_RELU | GF_DROPOUT | GF_RELU_BWD)))
          return (int)hipErrorInvalidValue;
        p.tiles_n = 1;
        tiles = (p.M + RM - 1) / RM;
      }
      if (akm) {
        if ((p.M & 7) || (p.N & 7) || p.M < 8 || p.N < 8 || !p.C || p.C16 || p.bias || p.res ||
            (p.flags & ~(GF_A_KMAJOR | GF_B_KMAJOR | GF_ATOMIC | GF_A_VIRT)))
          return (int)hipErrorInvalidValue;
        if (p.flags & GF_A_VIRT) {     // sample count rides in ldc16 (unused by TN); its gradient slice must fit the LDS spare
          const int ns = p.row_sample ? p.ldc16 : (p.uniform_n > 0 ? (p.K + p.uniform_n - 1) / p.uniform_n : 0);
          if (!p.virt_g || ns < 1 || ns * 256 > 2 * ROW_BUF_BYTES - 2 * BUF_BYTES || (!p.row_sample && p.uniform_n < 1))
            return (int)hipErrorInvalidValue;
        }
        const int ktiles = (p.K + 127) / 128 * 2;
        const int per = ktiles < kcap ? ktiles : kcap;                       // even
        p.kchunk = per * BK;
        p.ksplit = (ktiles + per - 1) / per;
      } else {
        if ((p.K % BK) || (p.N & 3) || (p.flags & (GF_ATOMIC | GF_SIGMOID)) || p.bias_grad) return (int)hipErrorInvalidValue;
        if ((p.C && (!al16(p.C) || (p.ldc & 3))) || (p.C16 && ((reinterpret_cast<uintptr_t>(p.C16) & 7) || (p.ldc16 & 3))) ||
            (p.bias && !al16(p.bias)) || (p.res && (!al16(p.res) || (p.ldr & 3))) || (p.colmean && !p.row_sample && p.uniform_n < 1))
          return (int)hipErrorInvalidValue;
        p.kchunk = p.K; p.ksplit = 1;
      }
      p.tile_begin = total;
      total += tiles * p.ksplit;
    }
    if (g_gemm16_tn_kcap > 0 || (kcap > 16 ? total >= 440 : total >= 200) || kcap <= KCAP_MIN) break;
  }
  gb.n_heavy = 0; gb.exp = g_gemm16_exp;
  // (measured on the training step: B = 4 -2.8 us, B = 16 +-0, B = 64 +0.7 us -- with long chunks the short blocks no longer
  // matter and the plain runs keep a K slice's tiles together; hence only up to 16-tile chunks)
  if (g_gemm16_balance && kcap_used <= 16) {                  // a prefix of split-K problems followed by single-slice ones only
    int i = 0;
    while (i < gb.n && (gb.p[i].flags & GF_A_KMAJOR) && gb.p[i].ksplit > 1) ++i;
    const int first_light = i;
    while (i < gb.n && (gb.p[i].flags & GF_A_KMAJOR) && gb.p[i].ksplit == 1) ++i;
    if (i == gb.n && first_light > 0 && first_light < gb.n) gb.n_heavy = gb.p[first_light].tile_begin;
  }
  bool has_tn = false, has_rows = false, has_virt = false;
  for (int i = 0; i < gb.n; ++i) {
    has_tn = has_tn || (gb.p[i].flags & GF_A_KMAJOR); has_rows = has_rows || gb.p[i].ln_mode;
    has_virt = has_virt || (gb.p[i].flags & GF_A_VIRT);
    if ((gb.p[i].flags & GF_A_VIRT) && !(gb.p[i].flags & GF_A_KMAJOR) && (!gb.p[i].ln_mode || !gb.p[i].virt_g || (gb.p[i].ldg & 3)))
      return (int)hipErrorInvalidValue;
  }
  if (has_virt && !has_rows) return (int)hipErrorInvalidValue;      // the virtual operand lives in the whole-row kernel only
  const size_t lds = has_rows ? 2 * ROW_BUF_BYTES : (has_tn ? 2 * BUF_BYTES : 2 * NT_BUF_BYTES);     // 72 / 64 / 48 KB
  static const bool attr_ok = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ROW_BUF_BYTES);
    return true;
  }();
  (void)attr_ok;
  double fl = 0.0;
  for (int i = 0; i < gb.n; ++i) fl += 2.0 * gb.p[i].M * (double)gb.p[i].N * gb.p[i].K;
  const int prof = gemm_prof_open(stream, fl);
  // two register stages (four measured slower on every launch of the step; round 2: a weight-gradient-only kernel with four
  // stages at two blocks per CU, 184 VGPRs, no spills: +1 us on the fused schedule's weight-gradient launch)
  if (has_rows) hipLaunchKernelGGL((gemm16_kernel<2, true>), dim3(total), dim3(256), lds, stream, gb, total);
  else          hipLaunchKernelGGL((gemm16_kernel<2, false>), dim3(total), dim3(256), lds, stream, gb, total);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}
