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

// Guarded loads, branch-free BY CONSTRUCTION.  hipcc turns "cond ? load : 0" into a branch around
// the load plus a vmcnt(0) wait per element (one memory round trip each; the first version of this
// kernel spent ~4 us per K step there).  So loading and masking are separate steps: raw_*() issues
// every load unconditionally from an address clamped into the matrix, the staged registers are
// pinned with an empty asm at the point of use, and only then mask_*() zeroes what lies outside.
struct Operand {
  const float* p; int ld; bool kmajor; bool vec;
  int outer;   // number of rows/cols (M or N)
  int K;
};

__device__ __forceinline__ void pin4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// VEC is a COMPILE-TIME switch: with a runtime (even wave-uniform) switch hipcc if-converts the two
// paths into four dword loads with selected addresses, i.e. never emits global_load_dwordx4.
// VEC requires ld % 4 == 0, a 16-B aligned base and the contiguous extent % 4 == 0.
// 4 consecutive k of one row/col (m-major source): element (r, k) at p[r*ld + k]
template <bool VEC>
__device__ __forceinline__ float4 raw_mm(const Operand& o, int r, int k) {
  const float* base = o.p + (size_t)min(r, o.outer - 1) * o.ld;
  if constexpr (VEC) {
    return *reinterpret_cast<const float4*>(base + min(k, o.K - 4));
  } else {
    const int km = o.K - 1;
    return make_float4(base[min(k, km)], base[min(k + 1, km)], base[min(k + 2, km)], base[min(k + 3, km)]);
  }
}
__device__ __forceinline__ float4 mask_mm(const Operand& o, float4 v, int r, int k, int kend) {
  const bool rok = r < o.outer;
  v.x = (rok && k < kend) ? v.x : 0.f;     v.y = (rok && k + 1 < kend) ? v.y : 0.f;
  v.z = (rok && k + 2 < kend) ? v.z : 0.f; v.w = (rok && k + 3 < kend) ? v.w : 0.f;
  return v;
}

// 4 consecutive rows/cols at one k (k-major source): element (m, k) at p[k*ld + m]
template <bool VEC>
__device__ __forceinline__ float4 raw_km(const Operand& o, int m, int k) {
  const float* base = o.p + (size_t)min(k, o.K - 1) * o.ld;
  if constexpr (VEC) {
    return *reinterpret_cast<const float4*>(base + min(m, o.outer - 4));
  } else {
    const int mm = o.outer - 1;
    return make_float4(base[min(m, mm)], base[min(m + 1, mm)], base[min(m + 2, mm)], base[min(m + 3, mm)]);
  }
}
__device__ __forceinline__ float4 mask_km(const Operand& o, float4 v, int m, int k, int kend) {
  const bool kok = k < kend;
  v.x = (kok && m < o.outer) ? v.x : 0.f;     v.y = (kok && m + 1 < o.outer) ? v.y : 0.f;
  v.z = (kok && m + 2 < o.outer) ? v.z : 0.f; v.w = (kok && m + 3 < o.outer) ? v.w : 0.f;
  return v;
}

// 4 consecutive k (stride ld) at one column of a k-major source: element (k, c) at p[k*ld + c]
__device__ __forceinline__ float4 raw_ks(const Operand& o, int c, int k) {
  const float* base = o.p + min(c, o.outer - 1);
  const int km = o.K - 1;
  return make_float4(base[(size_t)min(k, km) * o.ld], base[(size_t)min(k + 1, km) * o.ld],
                     base[(size_t)min(k + 2, km) * o.ld], base[(size_t)min(k + 3, km) * o.ld]);
}


// value -> C with the problem's fused epilogue (v already holds acc + bias)
__device__ __forceinline__ void epilogue_store(const GemmProb& P, const DropCfg& drop, float v, int row, int col) {
  const int flags = P.flags;
  if (flags & GF_RELU) v = fmaxf(v, 0.f);
  if ((flags & GF_DROPOUT) && drop.p > 0.f)
    v *= drop_mult(drop, P.drop_site, (uint32_t)row * (uint32_t)P.N + (uint32_t)col);
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

struct TileCtx {
  int m0, n0, tn, kbeg, kend;
};

// K loop shared by both precisions: tile t sits in LDS buffer t & 1; while it is multiplied, tile
// t+1 (loaded PIPE-1 iterations ago into stage (t+1) % PIPE) is converted into the other buffer and
// that stage is refilled with tile t+1+PIPE.  One barrier per tile.
// The body is STRAIGHT-LINE on purpose: every gload is unconditional (addresses are clamped into the
// block's K range, tiles past `nt` are masked to zero and multiply as zeros, the trip count is nt
// rounded up to PIPE).  With "if (more tiles) gload(...)" the refilled stage registers became PHIs at
// the join, hipcc materialised them as v_mov copies right after the loads, and the s_waitcnt vmcnt(0)
// in front of those copies drained the whole pipeline every K step (0.85 us per 32-deep tile).
#define PIPELINE_LOOP                                                                         \
  _Pragma("unroll") for (int j = 0; j < PIPE; ++j) gload(st[j], kbeg + j * BK);               \
  sstore(st[0], kbeg, 0);                                                                     \
  gload(st[0], kbeg + PIPE * BK);                                                             \
  __syncthreads();                                                                            \
  for (int t = 0; t < nt; t += PIPE) {                                                        \
    _Pragma("unroll") for (int j = 0; j < PIPE; ++j) {                                        \
      const int tt = t + j;                                                                   \
      sstore(st[(j + 1) % PIPE], kbeg + (tt + 1) * BK, (j + 1) & 1);                          \
      gload(st[(j + 1) % PIPE], kbeg + (tt + 1 + PIPE) * BK);                                 \
      compute(j & 1);                                                                         \
      __syncthreads();                                                                        \
    }                                                                                         \
  }

// Register stages of the global->LDS pipeline: PIPE (template parameter) K-tiles are in flight per block (their loads
// were issued PIPE-1 iterations before they are converted and written to LDS).  In a training step
// the operands were just written by the previous kernel, usually on another XCD, so they are served
// from the Infinity Cache / HBM (~1.5 us per dependent round trip): with one tile in flight the K
// loop ran at that latency per 32-deep step.
struct Stage { float4 a[2], b[4]; };

template <int PREC, int PIPE, bool akm, bool bkm, bool VEC>
__device__ __forceinline__ void gemm_tile_body(const GemmBatch& gb, const GemmProb& P, const TileCtx tc, char* smem_raw) {
  const int m0 = tc.m0, n0 = tc.n0, tn = tc.tn, kbeg = tc.kbeg, kend = tc.kend;
  const int M = P.M, N = P.N;
  // (K field = end of this block's K range: loads past it re-read its last rows instead of the next split's)
  const Operand oa{P.A, P.lda, akm, VEC, M, kend}, ob{P.B, P.ldb, bkm, VEC, N, kend};

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  // Accumulator orientation.  Plain-store problems (x.W^T, dy.W) feed the WEIGHT-side operand as MFMA "A",
  // so the tile is C^T: lane = output row m, registers = 4-column quads -> 16-B stores (dword stores made the
  // epilogue store-issue bound).  Weight gradients keep C orientation: their fp32 atomics run at full rate
  // only when a wave-instruction covers 128 contiguous bytes per row (lanes along n).
  constexpr bool TRANS = !(akm && bkm);

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float bsum = 0.f;   // bias-gradient partial (A k-major, first N tile)
  const bool do_bsum = (P.bias_grad != nullptr) && akm && (tn == 0);
  const int nt = (kend - kbeg + BK - 1) / BK;
  Stage st[PIPE];

  // thread -> staging slot maps.  f32: slot s = tid + 256 i; m-major (row s>>3, k4 s&7), k-major
  // (k s>>4 | s>>5, m4 s&15 | s&31).  bf16 k-major sources are transposed in registers, so there a
  // thread loads a 2(k) x 4(m) block of A and a 4(k) x 4(n) block of B.
  // bf16 k-major maps (see the LDS image note below): A thread = 2(k) x 4(m) block, B thread = 4(k) x 4(n) block.
  // Eight neighbouring lanes take eight neighbouring 4-row blocks (128 contiguous bytes of a source row), the next
  // lane bits walk the dwords of one 16-B chunk, the wave index picks the chunk.
  const int ka_tm = (tid & 7) | (((tid >> 5) & 1) << 3), ka_kp = ((tid >> 3) & 3) | ((tid >> 6) << 2);
  const int kb_tn = (tid & 7) | (((tid >> 4) & 3) << 3), kb_q = ((tid >> 3) & 1) | ((tid >> 6) << 1);
  auto gload = [&](Stage& r, int k0) {
    if constexpr (PREC == 0 || !akm) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int s = tid + 256 * i;
        r.a[i] = !akm ? raw_mm<VEC>(oa, m0 + (s >> 3), k0 + (s & 7) * 4) : raw_km<VEC>(oa, m0 + (s & 15) * 4, k0 + (s >> 4));
      }
    } else {
      const int kq = 2 * ka_kp, m = m0 + 4 * ka_tm;
      r.a[0] = raw_km<VEC>(oa, m, k0 + kq);
      r.a[1] = raw_km<VEC>(oa, m, k0 + kq + 1);
    }
    if constexpr (PREC == 0 || !bkm) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = tid + 256 * i;
        r.b[i] = !bkm ? raw_mm<VEC>(ob, n0 + (s >> 3), k0 + (s & 7) * 4) : raw_km<VEC>(ob, n0 + (s & 31) * 4, k0 + (s >> 5));
      }
    } else {
      const int kq = 4 * kb_q, n = n0 + 4 * kb_tn;
#pragma unroll
      for (int e = 0; e < 4; ++e) r.b[e] = raw_km<VEC>(ob, n, k0 + kq + e);
    }
  };
  // pin the staged registers (forces the loads above to stay unconditional), then mask
  auto mask_stage = [&](Stage& r, int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) pin4(r.a[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) pin4(r.b[i]);
    if constexpr (PREC == 0 || !akm) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int s = tid + 256 * i;
        r.a[i] = !akm ? mask_mm(oa, r.a[i], m0 + (s >> 3), k0 + (s & 7) * 4, kend) : mask_km(oa, r.a[i], m0 + (s & 15) * 4, k0 + (s >> 4), kend);
      }
    } else {
      const int kq = 2 * ka_kp, m = m0 + 4 * ka_tm;
      r.a[0] = mask_km(oa, r.a[0], m, k0 + kq, kend); r.a[1] = mask_km(oa, r.a[1], m, k0 + kq + 1, kend);
    }
    if constexpr (PREC == 0 || !bkm) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = tid + 256 * i;
        r.b[i] = !bkm ? mask_mm(ob, r.b[i], n0 + (s >> 3), k0 + (s & 7) * 4, kend) : mask_km(ob, r.b[i], n0 + (s & 31) * 4, k0 + (s >> 5), kend);
      }
    } else {
      const int kq = 4 * kb_q, n = n0 + 4 * kb_tn;
#pragma unroll
      for (int e = 0; e < 4; ++e) r.b[e] = mask_km(ob, r.b[e], n, k0 + kq + e, kend);
    }
  };

  if constexpr (PREC == 0) {
    float* Abuf = reinterpret_cast<float*>(smem_raw);               // [2][A_F32]
    float* Bbuf = Abuf + 2 * A_F32;                                  // [2][B_F32]
    auto sstore = [&](Stage& r, int k0, int buf) {
      mask_stage(r, k0);
      float* As = Abuf + buf * A_F32; float* Bs = Bbuf + buf * B_F32;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int s = tid + 256 * i;
        if (!akm) *reinterpret_cast<float4*>(As + (s >> 3) * LDM + (s & 7) * 4) = r.a[i];
        else      *reinterpret_cast<float4*>(As + (s >> 4) * LDKA + (s & 15) * 4) = r.a[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = tid + 256 * i;
        if (!bkm) *reinterpret_cast<float4*>(Bs + (s >> 3) * LDM + (s & 7) * 4) = r.b[i];
        else      *reinterpret_cast<float4*>(Bs + (s >> 5) * LDKB + (s & 31) * 4) = r.b[i];
      }
    };
    auto compute = [&](int buf) {
      const float* As = Abuf + buf * A_F32; const float* Bs = Bbuf + buf * B_F32;
      if (do_bsum && tid < BM) {
#pragma unroll 8
        for (int kk = 0; kk < BK; ++kk) bsum += As[kk * LDKA + tid];
      }
      // lane half h of MFMA step s feeds k = 16h + s; four steps per group keep the fragments small
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float a[4], b[2][4];
        if (!akm) {
          const float4 v = *reinterpret_cast<const float4*>(As + (wr * 32 + l31) * LDM + 16 * h + 4 * g);
          a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
        } else {
          const float* p = As + (16 * h + 4 * g) * LDKA + wr * 32 + l31;
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = p[e * LDKA];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = wc * 64 + j * 32 + l31;
          if (!bkm) {
            const float4 v = *reinterpret_cast<const float4*>(Bs + c * LDM + 16 * h + 4 * g);
            b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
          } else {
            const float* p = Bs + (16 * h + 4 * g) * LDKB + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) b[j][e] = p[e * LDKB];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (TRANS) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[0][e], a[e], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[1][e], a[e], acc[1], 0, 0, 0);
          } else {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[0][e], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[1][e], acc[1], 0, 0, 0);
          }
        }
      }
    };
    PIPELINE_LOOP
  } else {
    unsigned short* Abuf = reinterpret_cast<unsigned short*>(smem_raw);   // [2][A_BF16]
    unsigned short* Bbuf = Abuf + 2 * A_BF16;                              // [2][B_BF16]
    // bf16 image [row][k], 80-B rows.  m-major sources: a float4 is 4 consecutive k of one row -> one
    // 8-B store.  k-major sources: B 4(k) x 4(n) block -> four 8-B stores; A 2(k) x 4(m) -> four 4-B stores.
    // A thread's four stores go to rows 4t..4t+3 and 80-B rows put row r at bank 20r: blocks 4t, 4(t+2), ...
    // land on the same banks (the first layout was 8- and 16-way conflicted and the K loop of the weight
    // gradients ran at LDS-store speed).  So k-major images swap the four 16-B chunks of a row by
    // chunk ^= (row >> 3) & 3: with the thread maps above every 32-lane (b32) / 16-lane (b64) store group
    // covers 32 distinct banks, and the 8-lane groups of the b128 fragment reads still do.
    auto sstore = [&](Stage& r, int k0, int buf) {
      mask_stage(r, k0);
      unsigned short* As = Abuf + buf * A_BF16; unsigned short* Bs = Bbuf + buf * B_BF16;
      if (!akm) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int s = tid + 256 * i;
          *reinterpret_cast<uint2*>(As + (s >> 3) * LDH + (s & 7) * 4) = make_uint2(pack2(r.a[i].x, r.a[i].y), pack2(r.a[i].z, r.a[i].w));
        }
      } else {
        const int m = 4 * ka_tm;
        const int off = (((ka_kp >> 2) ^ ((ka_tm >> 1) & 3)) << 3) + 2 * (ka_kp & 3);   // swizzled chunk, dword in chunk
        *reinterpret_cast<uint32_t*>(As + (m + 0) * LDH + off) = pack2(r.a[0].x, r.a[1].x);
        *reinterpret_cast<uint32_t*>(As + (m + 1) * LDH + off) = pack2(r.a[0].y, r.a[1].y);
        *reinterpret_cast<uint32_t*>(As + (m + 2) * LDH + off) = pack2(r.a[0].z, r.a[1].z);
        *reinterpret_cast<uint32_t*>(As + (m + 3) * LDH + off) = pack2(r.a[0].w, r.a[1].w);
      }
      if (!bkm) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int s = tid + 256 * i;
          *reinterpret_cast<uint2*>(Bs + (s >> 3) * LDH + (s & 7) * 4) = make_uint2(pack2(r.b[i].x, r.b[i].y), pack2(r.b[i].z, r.b[i].w));
        }
      } else {
        const int n = 4 * kb_tn;
        const int kq = (((kb_q >> 1) ^ ((kb_tn >> 1) & 3)) << 3) + 4 * (kb_q & 1);      // swizzled chunk, half of it
        *reinterpret_cast<uint2*>(Bs + (n + 0) * LDH + kq) = make_uint2(pack2(r.b[0].x, r.b[1].x), pack2(r.b[2].x, r.b[3].x));
        *reinterpret_cast<uint2*>(Bs + (n + 1) * LDH + kq) = make_uint2(pack2(r.b[0].y, r.b[1].y), pack2(r.b[2].y, r.b[3].y));
        *reinterpret_cast<uint2*>(Bs + (n + 2) * LDH + kq) = make_uint2(pack2(r.b[0].z, r.b[1].z), pack2(r.b[2].z, r.b[3].z));
        *reinterpret_cast<uint2*>(Bs + (n + 3) * LDH + kq) = make_uint2(pack2(r.b[0].w, r.b[1].w), pack2(r.b[2].w, r.b[3].w));
      }
    };
    auto compute = [&](int buf) {
      const unsigned short* As = Abuf + buf * A_BF16; const unsigned short* Bs = Bbuf + buf * B_BF16;
      bf16x8 a[2], b[2][2];
      {
        const int row = wr * 32 + l31;
        const int c = akm ? (h ^ ((row >> 3) & 3)) : h;
        const unsigned short* p = As + row * LDH;
        a[0] = *reinterpret_cast<const bf16x8*>(p + 8 * c);
        a[1] = *reinterpret_cast<const bf16x8*>(p + 8 * (c ^ 2));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wc * 64 + j * 32 + l31;
        const int c = bkm ? (h ^ ((row >> 3) & 3)) : h;
        const unsigned short* p = Bs + row * LDH;
        b[j][0] = *reinterpret_cast<const bf16x8*>(p + 8 * c);
        b[j][1] = *reinterpret_cast<const bf16x8*>(p + 8 * (c ^ 2));
      }
      if (do_bsum && tid < BM) {
        // bias gradient from the bf16 image (what the MFMA sees), summed in fp32
        const unsigned short* p = As + tid * LDH;
#pragma unroll 8
        for (int kk = 0; kk < BK; ++kk) bsum += __uint_as_float((uint32_t)p[kk] << 16);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if constexpr (TRANS) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][s], a[s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1][s], a[s], acc[1], 0, 0, 0);
        } else {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[0][s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[1][s], acc[1], 0, 0, 0);
        }
      }
    };
    PIPELINE_LOOP
  }

  if (do_bsum && tid < BM && m0 + tid < M) atomicAdd(P.bias_grad + m0 + tid, bsum);

  // ---- epilogue, executed once per block but 32 elements deep per lane, so it is kept lean
  const int flags = P.flags;
  if constexpr (!TRANS) {
    // C orientation: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int rowb = m0 + wr * 32 + 4 * h;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wc * 64 + j * 32 + l31;
      if (col >= N) continue;
      const float bv = P.bias ? P.bias[col] : 0.f;
      float* cp = P.C + col;
      if (flags & GF_ATOMIC) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rowb + (r & 3) + 8 * (r >> 2);
          if (row < M) atomicAdd(cp + (size_t)row * P.ldc, acc[j][r] + bv);
        }
      } else {
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
          const int row = rowb + (r & 3) + 8 * (r >> 2);
          if (row < M) epilogue_store(P, gb.drop, acc[j][r] + bv, row, col);
        }
      }
    }
  } else {
    // C^T orientation: lane&31 = output row, registers 4g..4g+3 = columns c0..c0+3, c0 = 32j + 8g + 4(lane>>5)
    const int row = m0 + wr * 32 + l31;
    if (row < M) {
      const bool v4 = ((N & 3) == 0) && ((P.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.C) & 15) == 0) &&
                      (!P.bias || (reinterpret_cast<uintptr_t>(P.bias) & 15) == 0) &&
                      (!P.res || (((P.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.res) & 15) == 0)));
      const bool simple = !(flags & (GF_ATOMIC | GF_SIGMOID));
      if (v4 && simple) {
        const float floor_ = (flags & GF_RELU) ? 0.f : -INFINITY;
        const bool dodrop = (flags & GF_DROPOUT) && gb.drop.p > 0.f;
        float* crow = P.C + (size_t)row * P.ldc;
        // residual / aux row of this output row
        const float* rrow = nullptr; float rscale = 1.f;
        if (P.res) {
          if (flags & GF_RES_BCAST) {
            int sb;
            if (P.row_sample) { sb = P.row_sample[row]; rscale = P.inv_nr[sb]; }
            else { sb = row / P.uniform_n; rscale = 1.0f / (float)P.uniform_n; }
            rrow = P.res + (size_t)sb * P.ldr;
          } else {
            rrow = P.res + (size_t)row * P.ldr;
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c0 = n0 + wc * 64 + j * 32 + 8 * g + 4 * h;
            if (c0 < N) {
              float4 v = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
              if (P.bias) { const float4 bv = *reinterpret_cast<const float4*>(P.bias + c0); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
              v.x = fmaxf(v.x, floor_); v.y = fmaxf(v.y, floor_); v.z = fmaxf(v.z, floor_); v.w = fmaxf(v.w, floor_);
              if (dodrop) {
                const uint32_t idx = (uint32_t)row * (uint32_t)N + (uint32_t)c0;
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
              *reinterpret_cast<float4*>(crow + c0) = v;
            }
          }
      } else {
#pragma unroll 2
        for (int j = 0; j < 2; ++j)
#pragma unroll 4
          for (int r = 0; r < 16; ++r) {
            const int col = n0 + wc * 64 + j * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            if (col < N) epilogue_store(P, gb.drop, acc[j][r] + (P.bias ? P.bias[col] : 0.f), row, col);
          }
      }
    }
  }
}

// PIPE = 2 at two blocks per CU for launches with more tiles than the chip holds at once; PIPE = 4
// (one block per CU, ~170 VGPRs) for the launches of ~one tile per CU, which are pure latency.
template <int PREC, int PIPE>
__global__ __launch_bounds__(256, (PIPE <= 2 ? 2 : 1)) void gemm_grouped_kernel(const GemmBatch gb, int total_tiles) {
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
  TileCtx tc;
  tc.tn = t % P.tiles_n;
  tc.m0 = (t / P.tiles_n) * BM; tc.n0 = tc.tn * BN;
  tc.kbeg = ks * P.kchunk;
  tc.kend = min(P.K, tc.kbeg + P.kchunk);
  const bool akm = P.flags & GF_A_KMAJOR, bkm = P.flags & GF_B_KMAJOR;
  const bool vec = ((P.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.A) & 15) == 0) && (((akm ? P.M : P.K) & 3) == 0) &&
                   ((P.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.B) & 15) == 0) && (((bkm ? P.N : P.K) & 3) == 0);
  // block-uniform dispatch to a body compiled for this problem's operand layouts
  if (vec) {
    if (!akm && !bkm)      gemm_tile_body<PREC, PIPE, false, false, true>(gb, P, tc, smem_raw);
    else if (!akm && bkm)  gemm_tile_body<PREC, PIPE, false, true, true>(gb, P, tc, smem_raw);
    else if (akm && bkm)   gemm_tile_body<PREC, PIPE, true, true, true>(gb, P, tc, smem_raw);
    else                   gemm_tile_body<PREC, PIPE, true, false, true>(gb, P, tc, smem_raw);
  } else {
    if (!akm && !bkm)      gemm_tile_body<PREC, PIPE, false, false, false>(gb, P, tc, smem_raw);
    else if (!akm && bkm)  gemm_tile_body<PREC, PIPE, false, true, false>(gb, P, tc, smem_raw);
    else if (akm && bkm)   gemm_tile_body<PREC, PIPE, true, true, false>(gb, P, tc, smem_raw);
    else                   gemm_tile_body<PREC, PIPE, true, false, false>(gb, P, tc, smem_raw);
  }
}


// ---------------------------------------------------------------------------------------------
// Skinny problems (the per-sample "tail" of the model: M = batch size B <= 64 rows, or a weight
// gradient whose contraction runs over those B rows).  The 64x128 block tile above would spend
// its K loop on empty rows and serialise it (~1 us per 32-deep step); here the unit of work is a
// 16x16 output tile on v_mfma_f32_16x16x4_f32 (exact fp32), operands loaded straight from
// global memory (they are KBs and L2-resident):
//   * C = x.W^T / dy.W  (M <= 64): one block per tile, its 4 waves split K and combine through LDS;
//   * dW += dy^T.x      (K <= 64): one wave per tile, 4 tiles per block.
// Lane (x = lane & 15, q = lane >> 4) of MFMA step e within a 16-deep k block feeds k = 4q + e for
// both operands, so a lane's four steps come from one 16-B load where the source is k-contiguous.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One 16x16 tile (or, for dW problems, NW tiles: one per wave) of skinny problem P; `local` = tile index within
// the problem.  Every thread returns normally.
// Operand layouts and the 16-byte-load switch are COMPILE-TIME (block-uniform dispatch in skinny_tile below): with
// runtime switches the loads sat in branches, their destination registers became PHIs, and the s_waitcnt in front
// of the PHI copies drained the prefetch every k block (1.1 us per 16-deep block, one full memory round trip).
template <int NW, bool akm, bool bkm, bool VEC>
__device__ __forceinline__ void skinny_tile_body(const GemmProb& P, const DropCfg& drop, int local, float (*red)[4][64]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x = lane & 15, q = lane >> 4;
  const int M = P.M, N = P.N, K = P.K;
  const Operand oa{P.A, P.lda, akm, VEC, M, K}, ob{P.B, P.ldb, bkm, VEC, N, K};
  const int tiles_n = P.tiles_n;          // 16-wide column tiles
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int tile, kb0, kbstep;
  if (akm) { tile = local * NW + wave; kb0 = 0; kbstep = 1; }    // a wave per tile, whole K
  else     { tile = local; kb0 = wave; kbstep = NW; }            // a block per tile, waves interleave K
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * 16, n0 = tn * 16;
  const bool live = m0 < M;               // (only the a-wave-per-tile mode can run past the last tile)
  float bsum = 0.f;
  if (live) {
    const int nkb = (K + 15) >> 4;
    auto lda_ = [&](int kb) { const int k = kb * 16 + 4 * q; if constexpr (akm) return raw_ks(oa, m0 + x, k); else return raw_mm<VEC>(oa, m0 + x, k); };
    auto ldb_ = [&](int kb) { const int k = kb * 16 + 4 * q; if constexpr (bkm) return raw_ks(ob, n0 + x, k); else return raw_mm<VEC>(ob, n0 + x, k); };
    // D k blocks in flight in D statically indexed register stages (rotating the stages through moves would make
    // every move wait for its load); raw loads are address-clamped, blocks past K are masked to zero
    constexpr int D = 8;        // (K = 512 over 4 waves = 8 k blocks: the whole operand slice is requested up front)
    float4 sa[D], sb[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { sa[d] = lda_(kb0 + d * kbstep); sb[d] = ldb_(kb0 + d * kbstep); }
    for (int kb = kb0; kb < nkb; kb += D * kbstep) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int kk = kb + d * kbstep;
        pin4(sa[d]); pin4(sb[d]);
        const float4 a = mask_mm(oa, sa[d], m0 + x, kk * 16 + 4 * q, K);     // (row/col, k) validity is layout-independent
        const float4 b = mask_mm(ob, sb[d], n0 + x, kk * 16 + 4 * q, K);
        // refill the stage only after its old contents are dead (the fence keeps the loads below it): a refill
        // issued while they were still live went to other registers, and the copies back at the loop edge waited
        // for every load in flight
        asm volatile("" ::: "memory");
        sa[d] = lda_(kk + D * kbstep); sb[d] = ldb_(kk + D * kbstep);
        bsum += (a.x + a.y) + (a.z + a.w);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
      }
    }
  }
  bool active = live;
  if (!akm) {                              // (block-uniform)
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    active = wave == 0;
    if (active) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) v += red[w][r][lane];
        acc[r] += v;
      }
    }
  }
  if (!active) return;
  if (akm && P.bias_grad && tn == 0) {    // bias gradient: sum over k of A(m, k)
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (q == 0 && m0 + x < M) atomicAdd(P.bias_grad + m0 + x, bsum);
  }
  // 16x16 accumulator: col = lane & 15, row = 4*(lane >> 4) + reg
  const int col = n0 + x;
  if (col < N) {
    const float bv = P.bias ? P.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * q + r;
      if (row < M) epilogue_store(P, drop, acc[r] + bv, row, col);
    }
  }
}

template <int NW>
__device__ __forceinline__ void skinny_tile(const GemmProb& P, const DropCfg& drop, int local, float (*red)[4][64]) {
  const bool akm = P.flags & GF_A_KMAJOR, bkm = P.flags & GF_B_KMAJOR;
  const bool vec = (akm || (((P.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.A) & 15) == 0))) &&
                   (bkm || (((P.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.B) & 15) == 0))) && ((P.K & 3) == 0);
  if (vec) {
    if (!akm && !bkm)     skinny_tile_body<NW, false, false, true>(P, drop, local, red);
    else if (!akm && bkm) skinny_tile_body<NW, false, true, true>(P, drop, local, red);
    else if (akm && bkm)  skinny_tile_body<NW, true, true, true>(P, drop, local, red);
    else                  skinny_tile_body<NW, true, false, true>(P, drop, local, red);
  } else {
    if (!akm && !bkm)     skinny_tile_body<NW, false, false, false>(P, drop, local, red);
    else if (!akm && bkm) skinny_tile_body<NW, false, true, false>(P, drop, local, red);
    else if (akm && bkm)  skinny_tile_body<NW, true, true, false>(P, drop, local, red);
    else                  skinny_tile_body<NW, true, false, false>(P, drop, local, red);
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(const GemmBatch gb) {
  __shared__ float red[NW - 1][4][64];
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAXP; ++i)
    if (i < gb.n && (int)blockIdx.x >= gb.p[i].tile_begin) pi = i;
  skinny_tile<NW>(gb.p[pi], gb.drop, blockIdx.x - gb.p[pi].tile_begin, red);
}

}  // namespace

namespace {
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> ev;     // 2 per launch
  std::vector<double> flops;
  std::vector<int> kind;
  size_t used = 0;
  double kind_ms[PROF_KINDS] = {}, kind_fl[PROF_KINDS] = {};
  int kind_n[PROF_KINDS] = {};
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
  g_prof.flops.clear(); g_prof.kind.clear();
  g_prof.used = 0;
  g_prof.on = true;
  return 0;
}

int gemm_prof_end(double* total_ms, int* launches, double* total_flops) {
  g_prof.on = false;
  double ms = 0.0, fl = 0.0;
  for (int k = 0; k < PROF_KINDS; ++k) { g_prof.kind_ms[k] = 0.0; g_prof.kind_fl[k] = 0.0; g_prof.kind_n[k] = 0; }
  for (size_t i = 0; i < g_prof.used; ++i) {
    hipError_t r = hipEventSynchronize(g_prof.ev[2 * i + 1]);
    if (r != hipSuccess) return (int)r;
    float t = 0.f;
    r = hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
    if (r != hipSuccess) return (int)r;
    const int k = g_prof.kind[i];
    g_prof.kind_ms[k] += t; g_prof.kind_fl[k] += g_prof.flops[i]; g_prof.kind_n[k] += 1;
    ms += t; fl += g_prof.flops[i];
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = (int)g_prof.used;
  if (total_flops) *total_flops = fl;
  return 0;
}

int gemm_prof_kind(int kind, double* ms, int* launches, double* flops) {
  if (kind < 0 || kind >= PROF_KINDS) return (int)hipErrorInvalidValue;
  if (ms) *ms = g_prof.kind_ms[kind];
  if (launches) *launches = g_prof.kind_n[kind];
  if (flops) *flops = g_prof.kind_fl[kind];
  return 0;
}

int gemm_prof_open(hipStream_t stream, double flops, int kind) {
  if (!g_prof.on || 2 * (g_prof.used + 1) > g_prof.ev.size()) return -1;
  g_prof.flops.push_back(flops); g_prof.kind.push_back(kind < 0 || kind >= PROF_KINDS ? PROF_OTHER : kind);
  (void)hipEventRecord(g_prof.ev[2 * g_prof.used], stream);
  return (int)g_prof.used;
}
void gemm_prof_close(int slot, hipStream_t stream) {
  if (slot < 0) return;
  (void)hipEventRecord(g_prof.ev[2 * slot + 1], stream);
  ++g_prof.used;
}

static int launch_skinny(GemmBatch& gb, hipStream_t stream) {
  int total = 0;
  // (4 waves per block: 8- and 16-wave blocks for the long-K forward problems measured slower; with its K loop
  // removed a launch still takes 4.7 us -- the floor these launches sit on)
  const int nw = 4;
  for (int i = 0; i < gb.n; ++i) {
    GemmProb& p = gb.p[i];
    p.tiles_n = (p.N + 15) / 16;
    const int tiles = ((p.M + 15) / 16) * p.tiles_n;
    p.ksplit = 1; p.kchunk = p.K;
    p.tile_begin = total;
    total += (p.flags & GF_A_KMAJOR) ? (tiles + nw - 1) / nw : tiles;
  }
  if (total == 0) return 0;
  double fl = 0.0;
  for (int i = 0; i < gb.n; ++i) fl += 2.0 * gb.p[i].M * (double)gb.p[i].N * gb.p[i].K;
  const int prof = gemm_prof_open(stream, fl, PROF_TAIL);
  hipLaunchKernelGGL(gemm_skinny_kernel<4>, dim3(total), dim3(256), 0, stream, gb);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

int launch_gemm_batch(GemmBatch& gb, int precision, hipStream_t stream) {
  if (gb.n <= 0) return 0;
  if (precision == 0) {
    // per-sample tail: every problem is skinny (few rows, or a contraction over few rows)
    bool skinny = true;
    for (int i = 0; i < gb.n; ++i) {
      const GemmProb& p = gb.p[i];
      skinny = skinny && (((p.flags & GF_A_KMAJOR) ? p.K : p.M) <= 64);
    }
    if (skinny) return launch_skinny(gb, stream);
  }
  const int kcap = gb.kcap > 0 ? gb.kcap : 12;          // K tiles per split-K block of a weight-gradient problem (12: measured best on MI355X for long K)
  // tiles without split-K
  for (int i = 0; i < gb.n; ++i) {
    GemmProb& p = gb.p[i];
    p.tiles_n = (p.N + BN - 1) / BN;
  }
  int total = 0;
  for (int i = 0; i < gb.n; ++i) {
    GemmProb& p = gb.p[i];
    const int tiles = ((p.M + BM - 1) / BM) * p.tiles_n;
    int ksplit = 1;
    if ((p.flags & GF_ATOMIC) && (p.flags & GF_A_KMAJOR) && p.K > (gb.kcap > 0 ? kcap : 16) * BK) {
      // weight-gradient GEMM: small output, long contraction.  A block's K loop is serial
      // (~1 us per 32-deep tile), so cap it at ~12 tiles and let the fp32 atomics merge the splits.
      const int ktiles = (p.K + BK - 1) / BK;
      ksplit = (ktiles + kcap - 1) / kcap;
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
  const size_t lds = precision == 0 ? (size_t)2 * (A_F32 + B_F32) * 4 : (size_t)2 * (A_BF16 + B_BF16) * 2;
  int prof = -1;
  if (g_prof.on) {
    double fl = 0.0;
    for (int i = 0; i < gb.n; ++i) fl += 2.0 * gb.p[i].M * (double)gb.p[i].N * gb.p[i].K;
    prof = gemm_prof_open(stream, fl);
  }
  const bool deep = total <= 320;
  if (precision == 0) {
    if (deep) hipLaunchKernelGGL((gemm_grouped_kernel<0, 4>), dim3(total), dim3(256), lds, stream, gb, total);
    else      hipLaunchKernelGGL((gemm_grouped_kernel<0, 2>), dim3(total), dim3(256), lds, stream, gb, total);
  } else {
    if (deep) hipLaunchKernelGGL((gemm_grouped_kernel<1, 4>), dim3(total), dim3(256), lds, stream, gb, total);
    else      hipLaunchKernelGGL((gemm_grouped_kernel<1, 2>), dim3(total), dim3(256), lds, stream, gb, total);
  }
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}
