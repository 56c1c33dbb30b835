// The RG rows' whole forward in ONE launch, second design (gfx950): 64-row half-blocks of 4 waves, two independent blocks per CU.
// Contract and argument blocks: fused_rows.h (FrontStream, BackArgs: the same weight shadows, tile table, partial / ticket
// layout of the KG->RG attention and pooled-sum outputs as rgfwd_kernel of fused_wide.hip, which this replaces by size: inference calls
// from 10 240 packed rows, saving / dropout calls from 57 344); the KG rows' projections (Q2_16, KV16) come from the front kernel's KG
// launch as before, their chain (combine, out-projection, LayerNorm, FFN) runs as kgchain_kernel behind the RG launch.
//
// Why a second design.  rgfwd_kernel<4> runs one block of 8 waves per CU on 128 rows: the two
// waves of a SIMD execute the same program between the same barriers, so they reach their MFMA passes together and their
// epilogues together -- matrix time and vector time ADD (25 % MFMA issue), and the vector work itself was 9 instructions per
// MFMA (the MFMA gap hides 5-6).  Here
//   * a block is 4 waves (one per SIMD) on 64 rows, 79 KB of LDS: TWO blocks per CU that share nothing, so the two waves of a
//     SIMD belong to different blocks and drift through their phases independently -- one block's epilogue, barrier wait or
//     input load runs under the other block's MFMAs;
//   * a wave owns 2 feature tiles (= 2 attention heads) x 2 sub-tiles: a weight fragment feeds 2 MFMAs, an activation fragment
//     feeds 2 MFMAs (one LDS read and one L2 load per 2 MFMAs);
//   * the vector work per MFMA is cut by algebra, not by scheduling:
//       - biases enter as the first MFMA's C operand (a per-register constant vector read from LDS), never as adds;
//       - the KG->RG keys carry no bias at all (a constant per query cancels in its softmax) and the KG->RG values get theirs in
//         the combine (softmax weights sum to 1: O2 = sum p (v + b) = sum p v + b);
//       - invalid keys (Nk..15) are masked through the score MFMA's C operand (-1e30), not by selects;
//       - both softmaxes run in exp2 with log2(e) / sqrt(32) folded into one fma per score (the queries are not pre-scaled);
//       - LayerNorm statistics in one pass (sum and sum of squares: ONE cross-wave exchange instead of two);
//       - ReLU + bias as max(acc, -b), the bias re-added once per column sum;
//       - the mean pool of the LayerNorm output as MFMAs against an identity fragment (column sums fall out of the accumulator
//         layout: 16 in-lane adds and one cross-half add per tile instead of a 5-step lane butterfly per value).
//
// Numerics relative to rgfwd_kernel (both within north_star's 1e-3 of the f32 oracle; tests/test_hip_wide2.py): queries are
// rounded to bf16 unscaled; the pooled LayerNorm output is the sum of the bf16-rounded Y the FFN reads (fp32 accumulation).
#include "fused_rows.h"
#include "gemm.h"      // launch timing hooks
#include "wide2_inl.h"

namespace {


// LDS map (bytes).  The attention output lives in per-wave STRIPS [wave][rows][64 features of the wave's two heads] (pitch PS): a
// wave writes only its own strip, whose space it used before as scratch (the KG partial's Z, the transposing value reads), so
// nothing but the "attention output complete" barrier orders the waves there.  The input tile lives in the strips' space first.
// Constants: every bias / LayerNorm vector of the RG stream, once (no key bias: it cancels in the KG->RG softmax).
struct Cfg {
  static constexpr int BUFR = 0, TILE = ROWS * PR;                 // R tile, then (in place) the LayerNorm output Y
  static constexpr int BUFO = TILE, STRIP = ROWS * PS, STRIPS = NW * STRIP;
  static constexpr int XT = ROWS * PX, SCR = XT;                   // inside the strips' space until the attention output: input tile | per-wave scratch (4 KB each)
  static constexpr int RED = BUFO + STRIPS, RED_BYTES = NW * ROWS * 8;   // LayerNorm partials {sum, sum of squares} per (wave, row)
  static constexpr int CST = RED + RED_BYTES;
  // floats: b0 [256] | bq [256] | bv2 [256] (inference: the folded biases, launch_fold_rg) | bo [256] | ln_g [256] | ln_b [256] | b1 [512] |
  // bk2 [256] (training calls only: the saved keys carry their bias)
  static constexpr int C_B0 = 0, C_BQ = 256, C_BV = 512, C_BO = 768, C_G = 1024, C_BT = 1280, C_B1 = 1536, C_BK = 2048, C_FLOATS = 2304;
  static constexpr int LDS = CST + C_FLOATS * 4;
  // the KG rows' launch: one 32-row attention tile [32][256] (pitch PR) in the strips' space
  static_assert(XT + NW * 4096 <= STRIPS, "aliases of the strips' space");
  static_assert(LDS <= 81920, "two blocks per CU");
};

// ---- out-projection + residual -> LayerNorm -> FFN layer 0 (+ReLU) -> pooled sums, for RS sub-tiles.  The attention output (bf16)
// is read through `ofrag(s, ks)` (strips, or a plain [32][256] tile for the KG rows); the residual rows (bf16) sit in bufY, which
// the LayerNorm output overwrites in place (each wave reads and writes only its own 64 columns).  Wave w owns features
// 64 w .. + 63 of the 256-wide layers and 128 w .. + 127 of the FFN layer.  `cst`: bo | ln_g | ln_b | b1 at the offsets of Cfg.
// PAIR (the KG rows' launch): RS = 1 and the tile holds TWO samples of sub[0].nr <= 16 rows each -- sample sub[0].b in rows 0 .., sample
// sub[0].b + 1 (if `pair2`) in rows 16 .. -- pooled separately: accumulator registers 0..7 are rows 0..15, registers 8..15 rows 16..31.
// DROP / SAVE (training calls): dropout on the FFN activation (counter hash keyed by the element's packed index, common.h) and the saved
// set of the backward: normalised LayerNorm input XH16, 1 / std, LayerNorm output Y16, the ReLU-and-dropout bit mask.
// grow(r): tile row r (0 .. 32 RS - 1) -> its row in the stream's packed tensors, or -1.
// bufLo (SAVE): space for a second [32 RS][256] tile (pitch PR) that is free once the out-projection has read the attention output: the
// residual plane bf16(y - bf16(y)) of the LayerNorm output, so that the pooled mean of training calls is exact to 2^-17 (the per-sample
// tail's ReLU decisions sit behind it, and the tests bound how far a pre-activation may be from the oracle's: tests/helpers.py).
template <int RS, int DEPTH, bool PAIR, bool DROP, bool SAVE, class OF, class GR>
__device__ __forceinline__ void chain(const BackStream& S, int sbase, const Sub (&sub)[RS], OF&& ofrag, GR&& grow, const DropCfg& drop, char* bufY, char* bufLo, float* red, const float* cst,
                                      int w, int lane, Stage<RS, 2, 16, 2, DEPTH>& sto, unsigned long long* stamps, bool pair2 = false) {      // sbase: byte offset of S in the kernel's argument block (karg)
  static_assert(!PAIR || RS == 1, "pair mode: one tile");
  const int l31 = lane & 31, h = lane >> 5;
  const bool same = RS == 2 && sub[RS - 1].nr > 0 && sub[RS - 1].b == sub[0].b;      // (wave-uniform) both sub-tiles of one sample: one atomic per column
  // column sums of an accumulator tile in the lane = feature orientation (registers = rows), times 1 / n, into dst[sample][ld] + col
  auto pooled = [&](const f32x16 (&p)[RS], float add_per_row, float* dst, int ld) {
    if constexpr (PAIR) {
      float c0 = 0.f, c1 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { c0 += acc_row(i, h) < sub[0].nr ? p[0][i] + add_per_row : 0.f; c1 += acc_row(i, h) < sub[0].nr ? p[0][8 + i] + add_per_row : 0.f; }
      c0 = (c0 + __shfl_xor(c0, 32, 64)) * sub[0].inv_n; c1 = (c1 + __shfl_xor(c1, 32, 64)) * sub[0].inv_n;
      if (h == 0) { atomicAdd(dst + (size_t)sub[0].b * ld, c0); if (pair2) atomicAdd(dst + (size_t)(sub[0].b + 1) * ld, c1); }
    } else {
      float cs[RS];
#pragma unroll
      for (int s = 0; s < RS; ++s) {
        float c = 0.f;
        if (sub[s].nr >= 32) {                                    // (wave-uniform) full sub-tiles -- the common case -- carry no row masks
#pragma unroll
          for (int i = 0; i < 16; ++i) c += p[s][i];
          c = fmaf(16.f, add_per_row, c);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) c += acc_row(i, h) < sub[s].nr ? p[s][i] + add_per_row : 0.f;
        }
        cs[s] = (c + __shfl_xor(c, 32, 64)) * sub[s].inv_n;
      }
      if (same) {
        if (h == 0) atomicAdd(dst + (size_t)sub[0].b * ld, cs[0] + cs[RS - 1]);
      } else {
#pragma unroll
        for (int s = 0; s < RS; ++s)
          if (h == 0 && sub[s].nr > 0) atomicAdd(dst + (size_t)sub[s].b * ld, cs[s]);
      }
    }
  };
  Stage<RS, 2, 16, 4, DEPTH> stf;
  // identity fragment of the 32x32x16 MFMA: I[k][col] (B operand, lane = col) or I[row][k] (A operand, lane = row) -- the same registers:
  // lane (l31, h) holds k = 8 h + j of k step kk, 1.0 where 16 kk + 8 h + j == l31.  Two uses below: the residual rows enter the
  // out-projection's accumulators as I . R (two MFMAs per tile instead of unpacking and adding 16 bf16 values per lane), and the mean
  // pool of the LayerNorm output is Y . I.
  u32x4 idf[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int j = l31 - 16 * kk - 8 * h;                          // the element of this lane's fragment that is 1.0 (if 0 <= j < 8)
    const uint32_t one_lo = 0x3F80u, one_hi = 0x3F800000u;
    idf[kk] = u32x4{j == 0 ? one_lo : (j == 1 ? one_hi : 0u), j == 2 ? one_lo : (j == 3 ? one_hi : 0u),
                    j == 4 ? one_lo : (j == 5 ? one_hi : 0u), j == 6 ? one_lo : (j == 7 ? one_hi : 0u)};
  }
  {
    f32x16 acc[RS][2];
    {
      f32x16 init[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) init[t] = feature_vec(cst + Cfg::C_BO + 32 * (2 * w + t), h);
      sto.template run_f<true>(ofrag, init, acc);
    }
    // + residual: acc[feature][row] += sum_k I[feature][k] R[row][k] over the tile's own 32 features (exact: bf16 x 1.0 into fp32)
#pragma unroll
    for (int s = 0; s < RS; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const char* rp = bufY + (32 * s + l31) * PR + 2 * (32 * (2 * w + t)) + 16 * h;
        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(idf[0]), *reinterpret_cast<const bf16x8*>(rp), acc[s][t], 0, 0, 0);
        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(idf[1]), *reinterpret_cast<const bf16x8*>(rp + 32), acc[s][t], 0, 0, 0);
      }
    stamp(stamps, 9);
    // u = out-projection + bias (C operand) + residual; one-pass statistics over this lane's 32 features, then the other lane half
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { sm += acc[s][t][i]; sq = fmaf(acc[s][t][i], acc[s][t][i], sq); }
      sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
      if (h == 0) *reinterpret_cast<float2*>(red + 2 * (w * ROWS + 32 * s + l31)) = make_float2(sm, sq);
    }
    __syncthreads();
    stf.prefetch(LATEP(SAVE, const us16, sbase + KOFF(BackStream, W1), S.W1), w * (16 * 4), lane);                       // (flows during the normalisation and the pooling products)
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      float ts = 0.f, tq = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) {
        const float2 p = *reinterpret_cast<const float2*>(red + 2 * (ww * ROWS + 32 * s + l31));
        ts += p.x; tq += p.y;
      }
      const float mean = ts * (1.0f / 256.0f);
      const float var = fmaxf(tq * (1.0f / 256.0f) - mean * mean, 0.f);
      const float rstd = frsq(var + 1e-5f);
      const float nmr = -mean * rstd;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = 32 * (2 * w + t) + 8 * g + 4 * h;
          const f32x4 gm = *reinterpret_cast<const f32x4*>(cst + Cfg::C_G + c0), bt = *reinterpret_cast<const f32x4*>(cst + Cfg::C_BT + c0);
          const float x0 = fmaf(acc[s][t][4 * g], rstd, nmr), x1 = fmaf(acc[s][t][4 * g + 1], rstd, nmr);
          const float x2 = fmaf(acc[s][t][4 * g + 2], rstd, nmr), x3 = fmaf(acc[s][t][4 * g + 3], rstd, nmr);
          const float y0 = fmaf(x0, gm[0], bt[0]), y1 = fmaf(x1, gm[1], bt[1]), y2 = fmaf(x2, gm[2], bt[2]), y3 = fmaf(x3, gm[3], bt[3]);
          const u32x2 yh = u32x2{pack2(y0, y1), pack2(y2, y3)};
          *reinterpret_cast<u32x2*>(bufY + (32 * s + l31) * PR + 2 * c0) = yh;
          if constexpr (SAVE) {
            *reinterpret_cast<u32x2*>(bufLo + (32 * s + l31) * PR + 2 * c0) =
                u32x2{pack2(y0 - bf_lo(yh.x), y1 - bf_hi(yh.x)), pack2(y2 - bf_lo(yh.y), y3 - bf_hi(yh.y))};
          }
          if constexpr (SAVE) {                                 // normalised LayerNorm input, straight from the registers (8 bytes per row and lane half)
            const int gr = grow(32 * s + l31);
            if (gr >= 0) *reinterpret_cast<u32x2*>(karg<us16>(sbase + KOFF(BackStream, XH16)) + (size_t)gr * 256 + c0) = u32x2{pack2(x0, x1), pack2(x2, x3)};
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (SAVE) {
        const int gr = grow(32 * s + l31);
        if (w == 0 && h == 0 && gr >= 0) karg<float>(sbase + KOFF(BackStream, rstd))[gr] = rstd;
      }
    }
  }
  __syncthreads();                                              // Y tile complete
  stamp(stamps, 10);
  if constexpr (SAVE) {                                           // the LayerNorm output rows leave as 16-byte stores: 32 RS rows x 32 chunks
#pragma unroll 2
    for (int it = 0; it < RS * 4; ++it) {
      const int c = (int)threadIdx.x + NTH * it, r = c >> 5, k = c & 31;
      const int gr = grow(r);
      if (gr >= 0) *reinterpret_cast<u32x4*>(karg<us16>(sbase + KOFF(BackStream, Y16)) + (size_t)gr * 256 + 8 * k) = *reinterpret_cast<const u32x4*>(bufY + r * PR + 16 * k);
    }
  }
  // ---- mean pool of the LayerNorm output: column sums of the bf16 Y tile as MFMAs against an identity fragment.  Product
  // Y_s [32 rows][32 features of tile (2w + t)] . I: lane = feature, registers = rows -> 16 in-lane adds + the other lane half.
  {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 p[RS];
#pragma unroll
      for (int s = 0; s < RS; ++s) {
        const char* yp = bufY + (32 * s + l31) * PR + 2 * (32 * (2 * w + t)) + 16 * h;
        p[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(yp), as_frag(idf[0]), splat16(0.f), 0, 0, 0);
        p[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(yp + 32), as_frag(idf[1]), p[s], 0, 0, 0);
        if constexpr (SAVE) {
          const char* lp = bufLo + (32 * s + l31) * PR + 2 * (32 * (2 * w + t)) + 16 * h;
          p[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(lp), as_frag(idf[0]), p[s], 0, 0, 0);
          p[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(lp + 32), as_frag(idf[1]), p[s], 0, 0, 0);
        }
      }
      pooled(p, 0.f, LATEP(SAVE, float, sbase + KOFF(BackStream, Ymean), S.Ymean) + 32 * (2 * w + t) + l31, 256);
    }
  }
  // ---- FFN layer 0 + ReLU, pooled over the rows (lane = feature): two passes of 64 features per wave
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    f32x16 acc[RS][2];
    {
      f32x16 init[2] = {splat16(0.f), splat16(0.f)};
      stf.template run<false>(bufY + l31 * PR + 16 * h, 32 * PR, init, acc);
    }
    if (p == 0) stf.prefetch(LATEP(SAVE, const us16, sbase + KOFF(BackStream, W1), S.W1), w * (16 * 4) + 2, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int f = 128 * w + 64 * p + 32 * t + l31;
      const float bias = cst[Cfg::C_B1 + f], nb = -bias;
      f32x16 r[RS];
      if constexpr (!DROP && !SAVE) {
        // relu(a + b) = max(a, -b) + b: one max and one add per element, the bias re-added once per valid row
#pragma unroll
        for (int s = 0; s < RS; ++s)
#pragma unroll
          for (int i = 0; i < 16; ++i) r[s][i] = fmaxf(acc[s][t][i], nb);
        pooled(r, bias, LATEP(SAVE, float, sbase + KOFF(BackStream, Hmean), S.Hmean) + f, 512);
      } else {
#pragma unroll
        for (int s = 0; s < RS; ++s) {
          uint32_t wlo = 0u, whi = 0u;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            float v = fmaxf(acc[s][t][i] + bias, 0.f);
            if constexpr (DROP) {
              // element index = (the row's index in the stream's packed tensors) * 512 + f, as ONE per-lane base plus a constant per
              // register (rows past a tile's end draw a mask nobody reads) -- a grow() per element keeps 32 indices alive across both passes
              const uint32_t rbase = PAIR ? (uint32_t)((sub[0].b + (i >> 3)) * sub[0].nr + 4 * h) : (uint32_t)(sub[s].row0 + 4 * h);
              v *= drop_mult(drop, S.site_ffn, (rbase + (uint32_t)acc_row(PAIR ? (i & 7) : i, 0)) * 512u + (uint32_t)f);
            }
            r[s][i] = v;
            if constexpr (SAVE) {
              const unsigned long long bal = __ballot(v > 0.f);
              // (the ballot is a scalar pair: v_writelane drops each half into lane i directly -- a `lane == i` select per register
              // keeps 16 compare masks + 16 ballots = 64 SGPRs alive and was the training variants' main source of SGPR spills)
              SET_LANE(wlo, (uint32_t)bal, i);
              SET_LANE(whi, (uint32_t)(bal >> 32), i);
            }
          }
          if constexpr (SAVE) {                                 // lane i < 16 holds the words of tile rows acc_row(i, 0) and acc_row(i, 1) = + 4, features 32 (4 w + 2 p + t) ..
            if (lane < 16) {
              const int ra = 32 * s + acc_row(lane, 0);
              const int g0r = grow(ra), g1r = grow(ra + 4);
              uint32_t* mk = karg<uint32_t>(sbase + KOFF(BackStream, mask));
              if (g0r >= 0) mk[(size_t)g0r * 16 + 4 * w + 2 * p + t] = wlo;
              if (g1r >= 0) mk[(size_t)g1r * 16 + 4 * w + 2 * p + t] = whi;
            }
          }
        }
        pooled(r, 0.f, LATEP(SAVE, float, sbase + KOFF(BackStream, Hmean), S.Hmean) + f, 512);
      }
    }
  }
}

// Combine the partials {max, sum, Z} of the KG->RG attention of the block's two samples into their attention outputs (bf16: sample 0
// -> rows j < Nk of bufO, sample 1 -> rows 16 + j; the other rows stay zero) and add the values' bias (softmax weights sum to 1).
// Segment k of a sample whose first tile is t0 sits at tile slot k == 0 ? t0 : (t0 / RT + k) RT.  One pass with online rescaling,
// no tables, no barriers: a thread owns (head, query, 8 features) of both samples -- four independent items -- and keeps the loads
// of four segments of all of them in flight (the partials come from HBM / the Infinity Cache: 150 KB per sample, written by the
// previous launch on other XCDs -- what this costs is round trips and bytes, not arithmetic).
// bv2 == nullptr: the values already carry their bias (training calls: the saved values do, and dropped probabilities do not sum to 1);
// lse2 (training calls): the softmax {max, sum} per (sample, head, query), max in the scores' natural units -- the backward recomputes
// the probabilities from them.
template <bool SAVE>
__device__ __forceinline__ void kg_combine2(const BackArgs& a, const float* bv2, char* bufO, const int (&t0)[2], const int (&nseg)[2], int b0) {
  const int tid = threadIdx.x, Nk = a.Nk;
  const int hdlo = tid >> 6, j = (tid >> 2) & 15, f8 = tid & 3;
  const int jl = min(j, Nk - 1);                                  // (rows j >= Nk are not stored: those threads re-read row Nk - 1 and store nothing)
  float M[4], L[4];
  f32x4 acc[4][2];
#pragma unroll
  for (int it = 0; it < 4; ++it) { M[it] = -INFINITY; L[it] = 0.f; acc[it][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[it][1] = acc[it][0]; }
  const int nmax = max(nseg[0], nseg[1]);
  for (int s0 = 0; s0 < nmax; s0 += 4) {
    float mm[4][4], ll[4][4];
    u32x4 zz[4][4], zz1[SAVE ? 4 : 1][4];                        // (training: Z in fp32 -- two 16-byte loads per segment)
#pragma unroll
    for (int it = 0; it < 4; ++it) {                              // item = (sample it >> 1, head 4 (it & 1) + hdlo)
      const int smp = it >> 1, hd = 4 * (it & 1) + hdlo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sg = max(0, min(s0 + k, nseg[smp] - 1));
        const float* p = LATEP(SAVE, const float, KOFF(BackArgs, part), a.part)      /* (BackArgs leads KgChainArgs) */
             + ((size_t)(sg == 0 ? t0[smp] : (t0[smp] / RT + sg) * RT) * 8 + hd) * (SAVE ? PART_FLOATS_F32 : PART_FLOATS);
        mm[it][k] = p[jl]; ll[it][k] = p[16 + jl];
        if constexpr (SAVE) {
          zz[it][k] = *reinterpret_cast<const u32x4*>(p + 32 + jl * 32 + 8 * f8); zz1[it][k] = *reinterpret_cast<const u32x4*>(p + 32 + jl * 32 + 8 * f8 + 4);
        } else {
          zz[it][k] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const us16*>(p + 32) + jl * 32 + 8 * f8);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int smp = it >> 1;
      float Mn = M[it];
#pragma unroll
      for (int k = 0; k < 4; ++k) Mn = s0 + k < nseg[smp] ? fmaxf(Mn, mm[it][k]) : Mn;
      const float r = fexp2(M[it] - Mn);                          // (maxima are kept in log2 units; first chunk: exp2(-inf) = 0)
      L[it] *= r; acc[it][0] *= r; acc[it][1] *= r;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (s0 + k < nseg[smp]) {
          const float e = fexp2(mm[it][k] - Mn);
          L[it] = fmaf(ll[it][k], e, L[it]);
          const u32x4 z = zz[it][k];
          if constexpr (SAVE) {
            acc[it][0] += __builtin_bit_cast(f32x4, z) * e; acc[it][1] += __builtin_bit_cast(f32x4, zz1[it][k]) * e;
          } else {
            acc[it][0] += f32x4{bf_lo(z[0]), bf_hi(z[0]), bf_lo(z[1]), bf_hi(z[1])} * e;
            acc[it][1] += f32x4{bf_lo(z[2]), bf_hi(z[2]), bf_lo(z[3]), bf_hi(z[3])} * e;
          }
        }
      M[it] = Mn;
    }
  }
  if (j < Nk) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int smp = it >> 1, hd = 4 * (it & 1) + hdlo;
      if (nseg[smp] <= 0) continue;                               // (block-uniform: no second sample)
      const float il = 1.0f / L[it];
      f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      if (bv2) { c0 = *reinterpret_cast<const f32x4*>(bv2 + 32 * hd + 8 * f8); c1 = *reinterpret_cast<const f32x4*>(bv2 + 32 * hd + 8 * f8 + 4); }
      *reinterpret_cast<u32x4*>(bufO + (16 * smp + j) * PR + 2 * (32 * hd + 8 * f8)) =
          u32x4{pack2(fmaf(acc[it][0][0], il, c0[0]), fmaf(acc[it][0][1], il, c0[1])), pack2(fmaf(acc[it][0][2], il, c0[2]), fmaf(acc[it][0][3], il, c0[3])),
                pack2(fmaf(acc[it][1][0], il, c1[0]), fmaf(acc[it][1][1], il, c1[1])), pack2(fmaf(acc[it][1][2], il, c1[2]), fmaf(acc[it][1][3], il, c1[3]))};
      if constexpr (SAVE) {
        if (f8 == 0 && a.lse2) { float* o = a.lse2 + (((size_t)(b0 + smp) * 8 + hd) * 16 + j) * 2; o[0] = M[it] * (1.0f / LOG2E); o[1] = L[it]; }
      }
    }
  }
}

// ---- the KG rows' launch (behind the RG rows'): per block TWO samples -- their 2 x Nk <= 32 KG rows are one 32-row tile -- combine each
// sample's KG->RG partials into its attention output, then out-projection + residual -> LayerNorm -> FFN layer 0 -> pooled sums.
// Why its own launch and not the RG block that completes a sample (fused_wide.hip does that, and so did this file's first build): the
// extra 19 us made one RG block in eight 65 % longer than the rest, and the dispatcher deals blocks to XCDs / shader engines in a fixed
// rotation -- with unequal blocks a fifth of the CU slots sat empty behind the long ones (measured: 400 of 512 slots busy at B = 1024;
// before the units were spread evenly over the XCDs, one XCD drew 854 of 987 chains and ran 230 us behind the other seven).  With
// every RG block alike the slots stay full, the partials need no write-through stores, no tickets and no acquire, and the KG rows of
// two samples share one weight stream.
struct KgChainArgs { BackArgs b; const float* bv2; };      // bv2: the KG->RG values' bias the combine adds (inference calls), null when the values carry it

template <int DEPTH, bool DROP, bool SAVE>
__global__ __launch_bounds__(NTH, 2) void kgchain_kernel(const KgChainArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const BackArgs& a = g.b;
  const BackStream& K = a.s[1];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, Nk = a.Nk;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* kbufY = smem + Cfg::BUFR; char* kbufO = smem + Cfg::BUFO;
  float* red = reinterpret_cast<float*>(smem + Cfg::RED);
  float* cst = reinterpret_cast<float*>(smem + Cfg::CST);
  const int b0 = 2 * (int)blockIdx.x;
  const bool pair2 = b0 + 1 < a.B;                                  // (block-uniform)
  stamp(a.stamps, 0);
  const int ta0 = a.tile_off[b0], ta1 = a.tile_off[b0 + 1], tb1 = pair2 ? a.tile_off[b0 + 2] : ta1;
  Stage<1, 2, 16, 2, DEPTH> stk;
  stk.prefetch(LATEP(SAVE, const us16, KOFF(KgChainArgs, b.s[1].Wo), K.Wo), w * (16 * 2), lane);
  // constants: the KG->RG values' bias (added by the combine) | bo | ln_g | ln_b | b1 of the KG stream
  for (int i = tid; i < 384; i += NTH) {
    if (i < 64 && !g.bv2) continue;
    const float* src = i < 64 ? g.bv2 + 4 * i : (i < 128 ? K.bo + 4 * (i - 64) : (i < 192 ? K.ln_g + 4 * (i - 128) : (i < 256 ? K.ln_b + 4 * (i - 192) : K.b1 + 4 * (i - 256))));
    *reinterpret_cast<float4*>(cst + (i < 64 ? Cfg::C_BV : Cfg::C_BO - 256) + 4 * i) = *reinterpret_cast<const float4*>(src);
  }
  // attention tile cleared (rows past Nk of either sample stay zero); residual rows G: sample b0 -> rows 0 .., sample b0 + 1 -> rows 16 ..
  for (int c = tid; c < 32 * PR / 16; c += NTH) reinterpret_cast<u32x4*>(kbufO)[c] = u32x4{0u, 0u, 0u, 0u};
  // tile row r -> row of the KG stream's packed tensors (or -1)
  auto grow = [&](int r) { const int smp = r >> 4, j = r & 15; return (j < Nk && (smp == 0 || pair2)) ? (b0 + smp) * Nk + j : -1; };
  for (int c = tid; c < 32 * 32; c += NTH) {
    const int r = c >> 5, k = c & 31, gr = grow(r);
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (gr >= 0) v = *reinterpret_cast<const u32x4*>(LATEP(SAVE, const us16, KOFF(KgChainArgs, b.s[1].R16), K.R16) + (size_t)gr * 256 + 8 * k);
    *reinterpret_cast<u32x4*>(kbufY + r * PR + 16 * k) = v;
  }
  __syncthreads();
  {
    const int t0[2] = {ta0, ta1};
    const int ns[2] = {(ta1 - 1) / RT - ta0 / RT + 1, pair2 ? (tb1 - 1) / RT - ta1 / RT + 1 : 0};
    kg_combine2<SAVE>(a, g.bv2 ? cst + Cfg::C_BV : nullptr, kbufO, t0, ns, b0);
  }
  __syncthreads();
  stamp(a.stamps, 13);
  if constexpr (SAVE) {                                           // the attention output rows (bf16): 32 rows x 32 chunks of 16 bytes
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = tid + NTH * it, r = c >> 5, k = c & 31, gr = grow(r);
      if (gr >= 0) *reinterpret_cast<u32x4*>(karg<us16>(KOFF(KgChainArgs, b.s[1].O16)) + (size_t)gr * 256 + 8 * k) = *reinterpret_cast<const u32x4*>(kbufO + r * PR + 16 * k);
    }
  }
  const Sub ksub[1] = {Sub{b0, b0 * Nk, Nk, 1.0f / (float)Nk}};
  chain<1, DEPTH, true, DROP, SAVE>(K, KOFF(KgChainArgs, b.s[1]), ksub, [&](int, int ks) { return *reinterpret_cast<const bf16x8*>(kbufO + l31 * PR + 32 * ks + 16 * h); }, grow, a.drop,
                                    kbufY, kbufO, red, cst, w, lane, stk, nullptr, pair2);
  stamp(a.stamps, 12);
}

// ---- the folded in-projection of the RG rows: [q | k2 | v2] = R Win^T + bin with R = x Wrg^T + brg  ==  x Wf^T + bf,
//   Wf = Win Wrg  [768 x 128],   bf = Win brg + bin  [768],   Win = [Wq1; Wk2; Wv2] (fusion_model.py:108-117, 123-126).
// One launch per parameter change (an inference loop keeps the result with its weight shadows): fp32 products, Wf leaves as the bf16
// shadow in fragment order ([wave][k step][tile][lane][8], fused_rows.h), bf as fp32.  Thread = one 16-byte shadow chunk (8 k of one row).
__global__ __launch_bounds__(256) void fold_rg_kernel(const float* __restrict__ Wq, const float* __restrict__ Wkv, const float* __restrict__ bq, const float* __restrict__ bkv,
                                                      const float* __restrict__ Wrg, const float* __restrict__ brg, us16* __restrict__ Wf, float* __restrict__ bf) {
  const int c = (int)blockIdx.x * 256 + (int)threadIdx.x;
  constexpr int KS = 8, NTW = 6, CHUNKS = 768 * 128 / 8;
  if (c < CHUNKS) {
    const int lane = c & 63;
    int r = c >> 6;
    const int t = r % NTW; r /= NTW;
    const int ks = r % KS, w = r / KS;
    const int n = 32 * (w * NTW + t) + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    const float* wrow = n < 256 ? Wq + (size_t)n * 256 : Wkv + (size_t)(n - 256) * 256;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int j = 0; j < 256; j += 4) {
      const float4 a = *reinterpret_cast<const float4*>(wrow + j);
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 x0 = *reinterpret_cast<const float4*>(Wrg + (size_t)(j + q) * 128 + k0), x1 = *reinterpret_cast<const float4*>(Wrg + (size_t)(j + q) * 128 + k0 + 4);
        acc[0] = fmaf(av[q], x0.x, acc[0]); acc[1] = fmaf(av[q], x0.y, acc[1]); acc[2] = fmaf(av[q], x0.z, acc[2]); acc[3] = fmaf(av[q], x0.w, acc[3]);
        acc[4] = fmaf(av[q], x1.x, acc[4]); acc[5] = fmaf(av[q], x1.y, acc[5]); acc[6] = fmaf(av[q], x1.z, acc[6]); acc[7] = fmaf(av[q], x1.w, acc[7]);
      }
    }
    reinterpret_cast<u32x4*>(Wf)[c] = u32x4{pack2(acc[0], acc[1]), pack2(acc[2], acc[3]), pack2(acc[4], acc[5]), pack2(acc[6], acc[7])};
  } else if (c < CHUNKS + 768) {
    const int n = c - CHUNKS;
    const float* wrow = n < 256 ? Wq + (size_t)n * 256 : Wkv + (size_t)(n - 256) * 256;
    float acc = n < 256 ? bq[n] : bkv[n - 256];
    for (int j = 0; j < 256; ++j) acc = fmaf(wrow[j], brg[j], acc);
    bf[n] = acc;
  }
}

struct RgFwd2Args { FrontStream f; BackArgs b; float qscale; const us16* Wf; const float* bf; int save_r16; };      // Wf / bf: the folded in-projection (launch_fold_rg)

// FOLD: the in-projections read the input tile through the folded weights (inference calls); otherwise the R tile through the
// unfolded [768 x 256] shadow, bit for bit what the backward kernels and the bf16-operand oracle assume.  DROP / SAVE (training
// calls): dropout at the three sites of this kernel and the saved set of the backward (X16, R16, Q16 pre-scaled, K2 | V2 with their
// biases, O16, XH16, 1 / std, Y16, the ReLU-and-dropout mask); the KG rows' launch saves its own half.
template <int DEPTH, bool FOLD, bool DROP, bool SAVE>
__global__ __launch_bounds__(NTH, 2) void rgfwd2_kernel(const RgFwd2Args g) {
  static_assert(!(FOLD && SAVE), "the backward assumes the unfolded in-projection");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FrontStream& F = g.f;
  const BackArgs& a = g.b;
  const BackStream& S = a.s[0];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, Nk = a.Nk;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* bufR = smem + Cfg::BUFR; char* strips = smem + Cfg::BUFO; char* bufX = strips;
  char* strip = strips + w * Cfg::STRIP;                          // this wave's strip (the attention output)
  // this wave's scratch: beside the input tile while the folded passes still read it; otherwise the wave's own strip (free until the
  // attention output; training calls also stage their saved keys / queries there for 16-byte row stores)
  char* scr = FOLD ? strips + Cfg::SCR + w * 4096 : strip;
  float* red = reinterpret_cast<float*>(smem + Cfg::RED);
  float* cst = reinterpret_cast<float*>(smem + Cfg::CST);
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
  const bool same = sub[1].nr > 0 && sub[1].b == sub[0].b;        // (wave-uniform) both sub-tiles belong to one sample
  Stage<RT, 2, 8, 2, DEPTH> st0;
  st0.prefetch(LATEP(SAVE, const us16, KOFF(RgFwd2Args, f.W0), F.W0), w * (8 * 2), lane);
  // ---- input rows: fp32 -> bf16 tile (rows past a sub-tile's end cleared)
  {
    constexpr int XIT = ROWS * 8 / NTH;
    float4 xv[XIT][4];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      // (XIT = 2: iteration `it` covers exactly sub-tile `it`'s 32 rows, so the row base is a scalar and the lane part a 32-bit offset)
      static_assert(ROWS * 8 / NTH == RT, "one sub-tile per iteration");
      const int r = tid >> 3, c = tid & 7;
      const float* xbase = F.X + (size_t)sub[it].row0 * 128;
      const unsigned xoff = (unsigned)max(0, min(r, sub[it].nr - 1)) * 128u + 16u * (unsigned)c;
      const float4* src = reinterpret_cast<const float4*>(xbase + xoff);
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[it][q] = src[q];
    }
    // every bias / LayerNorm vector of the RG stream -> LDS, once: 512 float4 (folded biases: the queries' and the values' thirds)
#pragma unroll
    for (int i = tid; i < Cfg::C_FLOATS / 4; i += NTH) {
      const float* bqs = FOLD ? g.bf : F.bq;                       // queries' bias, values' bias (folded or plain)
      const float* bvs = FOLD ? g.bf + 512 : F.bkv + 256;
      const float* src = i < 64 ? F.b0 + 4 * i : (i < 128 ? bqs + 4 * (i - 64) : (i < 192 ? bvs + 4 * (i - 128) : (i < 256 ? S.bo + 4 * (i - 192) :
                         (i < 320 ? S.ln_g + 4 * (i - 256) : (i < 384 ? S.ln_b + 4 * (i - 320) : S.b1 + 4 * (i - 384))))));
      *reinterpret_cast<float4*>(cst + 4 * i) = *reinterpret_cast<const float4*>(src);
    }
    if constexpr (!FOLD) {                                        // the keys' bias (the saved keys carry it)
      if (tid < 64) *reinterpret_cast<float4*>(cst + Cfg::C_BK + 4 * tid) = *reinterpret_cast<const float4*>(F.bkv + 4 * tid);
    }
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int r = 32 * it + (tid >> 3), c = tid & 7;
      const bool ok = (tid >> 3) < sub[it].nr;
      const float4 v0 = xv[it][0], v1 = xv[it][1], v2 = xv[it][2], v3 = xv[it][3];
      u32x4 p0 = u32x4{pack2(v0.x, v0.y), pack2(v0.z, v0.w), pack2(v1.x, v1.y), pack2(v1.z, v1.w)};
      u32x4 p1 = u32x4{pack2(v2.x, v2.y), pack2(v2.z, v2.w), pack2(v3.x, v3.y), pack2(v3.z, v3.w)};
      if (!ok) { p0 = u32x4{0u, 0u, 0u, 0u}; p1 = p0; }
      *reinterpret_cast<u32x4*>(bufX + r * PX + 32 * c) = p0;
      *reinterpret_cast<u32x4*>(bufX + r * PX + 32 * c + 16) = p1;
      if constexpr (SAVE) {
        if (ok) { u32x4* d = reinterpret_cast<u32x4*>(karg<us16>(KOFF(RgFwd2Args, f.X16)) + ((size_t)sub[it].row0 + (tid >> 3)) * 128 + 16 * c); d[0] = p0; d[1] = p1; }
      }
    }
  }
  __syncthreads();
  stamp(a.stamps, 1);
  // tile row r (0 .. 63) -> packed RG row, or -1 past its sub-tile's end
  auto grow = [&](int r) { const int rr = r & 31; return r < 32 ? (rr < sub[0].nr ? sub[0].row0 + rr : -1) : (rr < sub[1].nr ? sub[1].row0 + rr : -1); };
  // ---- projection 128 -> 256: wave w owns features 64 w .. + 63 of the R tile
  // the in-projections read the INPUT tile: [q | k2 | v2] = x Wf^T + bf with Wf = [Wq; Wk2; Wv2] Wrg (768 x 128: half the k steps and
  // half the weight bytes of the unfolded 256 -> 768 product; launch_fold_rg)
  constexpr int KS1 = FOLD ? 8 : 16;                               // k steps of an in-projection pass: over the input tile (128) or the R tile (256)
  Stage<RT, 2, KS1, 6, DEPTH> st1;
  auto W1s = [&]() -> const us16* { return FOLD ? g.Wf : LATEP(SAVE, const us16, KOFF(RgFwd2Args, f.W1), F.W1); };
  const char* act1 = FOLD ? bufX + l31 * PX + 16 * h : bufR + l31 * PR + 16 * h;
  constexpr int sub1 = FOLD ? 32 * PX : 32 * PR;
  auto w1pair = [&](int tg) { return (tg / 6) * (KS1 * 6) + tg % 6; };    // first fragment of tiles tg, tg + 1 (tg even) in the [768 x 128 | 256] shadow
  {
    f32x16 acc[RT][2];
    {
      f32x16 init[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) init[t] = feature_vec(cst + Cfg::C_B0 + 32 * (2 * w + t), h);
      st0.template run<true>(bufX + l31 * PX + 16 * h, 32 * PX, init, acc);
    }
    st1.prefetch(W1s(), w1pair(8 + 2 * w), lane);
#pragma unroll
    for (int s = 0; s < RT; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *reinterpret_cast<u32x2*>(bufR + (32 * s + l31) * PR + 2 * (32 * (2 * w + t) + 8 * gq + 4 * h)) =
              u32x2{pack2(acc[s][t][4 * gq], acc[s][t][4 * gq + 1]), pack2(acc[s][t][4 * gq + 2], acc[s][t][4 * gq + 3])};
  }
  __syncthreads();                                                // R tile complete
  stamp(a.stamps, 2);
  if (SAVE && g.save_r16) {                                       // R16 (read by the row-space weight gradients only): the tile's rows as 16-byte stores
#pragma unroll 2
    for (int it = 0; it < RT * 4; ++it) {                         // (two in flight at a time: fully unrolled, hipcc hoists all eight reads over the pass's own prefetches)
      const int c = tid + NTH * it, r = c >> 5, k = c & 31, gr = grow(r);
      if (gr >= 0) *reinterpret_cast<u32x4*>(karg<us16>(KOFF(RgFwd2Args, f.R16)) + (size_t)gr * 256 + 8 * k) = *reinterpret_cast<const u32x4*>(bufR + r * PR + 16 * k);
    }
  }
  // RG->KG scores -> log2 units: the folded queries are NOT pre-scaled; the unfolded ones are (and are saved so)
  const float sc2 = FOLD ? LOG2E * g.qscale : LOG2E;
  // ---- passes k2, v2 (heads 2w, 2w + 1): the KG->RG partial of this block's rows
  {
    u32x4 ef[RT][2][2];
    float Lsub[RT][2], mseg[RT][2];
    {
      // the samples' KG queries in accumulator k order (B operand: lane = query).  They arrive PRE-SCALED by 1 / sqrt(32) (the KG
      // front launch): the scores below need log2(e) only.
      u32x4 q2f[RT][2][2];
#pragma unroll
      for (int s = 0; s < RT; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const us16* q2p = (LATEP(SAVE, const us16, KOFF(RgFwd2Args, b.Q2_16), a.Q2_16) + ((size_t)sub[s].b * Nk * 256 + 32 * (2 * w + t))) + ((unsigned)min(l31, Nk - 1) * 256u + 4u * (unsigned)h);      // (scalar base + 32-bit lane offset)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const u32x2 lo = *reinterpret_cast<const u32x2*>(q2p + 16 * kk), hi = *reinterpret_cast<const u32x2*>(q2p + 16 * kk + 8);
            q2f[s][t][kk] = u32x4{lo.x, lo.y, hi.x, hi.y};
          }
        }
      f32x16 acc[RT][2];
      {
        // (inference: no key bias -- a constant per query cancels in its softmax; training: the saved keys carry it)
        f32x16 init[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) init[t] = FOLD ? splat16(0.f) : feature_vec(cst + Cfg::C_BK + 32 * (2 * w + t), h);
        st1.template run<true>(act1, sub1, init, acc);
      }
      st1.prefetch(W1s(), w1pair(16 + 2 * w), lane);
      stamp(a.stamps, 3);
      // one head (feature tile t) at a time: scores of both sub-tiles, their maxima (a run of one sample shares its maximum), the
      // exponentials -- 32 score registers live instead of 64
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 S2[RT];
        float mx[RT];
#pragma unroll
        for (int s = 0; s < RT; ++s) {
          const u32x4 k0 = pack8(acc[s][t], 0), k1 = pack8(acc[s][t], 1);
          if constexpr (SAVE) {                                   // K2 -> the wave's strip [row][64 features]: lane = row; features 16 kk + 4 h .. + 3 and 16 kk + 8 + 4 h .. + 3 of head t
            char* kd = strip + (32 * s + l31) * PS + 2 * (32 * t + 4 * h);
            *reinterpret_cast<u32x2*>(kd) = u32x2{k0.x, k0.y}; *reinterpret_cast<u32x2*>(kd + 16) = u32x2{k0.z, k0.w};
            *reinterpret_cast<u32x2*>(kd + 32) = u32x2{k1.x, k1.y}; *reinterpret_cast<u32x2*>(kd + 48) = u32x2{k1.z, k1.w};
          }
          S2[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(k0), as_frag(q2f[s][t][0]), splat16(0.f), 0, 0, 0);   // [key row][query]: lane = query
          S2[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(k1), as_frag(q2f[s][t][1]), S2[s], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < RT; ++s) {
          float m = -INFINITY;
          if (sub[s].nr >= 32) {
#pragma unroll
            for (int i = 0; i < 16; ++i) m = fmaxf(m, S2[s][i]);
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) m = acc_row(i, h) < sub[s].nr ? fmaxf(m, S2[s][i]) : m;
          }
          mx[s] = fmaxf(m, __shfl_xor(m, 32, 64));
        }
        mseg[0][t] = same ? fmaxf(mx[0], mx[1]) : mx[0];
        mseg[1][t] = same ? mseg[0][t] : mx[1];
#pragma unroll
        for (int s = 0; s < RT; ++s) {
          const float nm = -mseg[s][t] * LOG2E;
          float e[16], L = 0.f;
          if (sub[s].nr >= 32) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { e[i] = fexp2(fmaf(S2[s][i], LOG2E, nm)); L += e[i]; }
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) { e[i] = acc_row(i, h) < sub[s].nr ? fexp2(fmaf(S2[s][i], LOG2E, nm)) : 0.f; L += e[i]; }
          }
          Lsub[s][t] = L;
          if constexpr (DROP) {                                   // (the row sums stay undropped: they normalise the probabilities, which are dropped afterwards)
#pragma unroll
            for (int i = 0; i < 16; ++i)
              e[i] *= drop_mult(a.drop, SITE_ATTN_KG2RG, (((uint32_t)(sub[s].row0 + 4 * h) * 8u + (uint32_t)(2 * w + t)) * (uint32_t)Nk + (uint32_t)l31) +
                                                         (uint32_t)(8 * acc_row(i, 0)) * (uint32_t)Nk);
          }
#pragma unroll
          for (int k = 0; k < 2; ++k)
            ef[s][t][k] = u32x4{pack2(e[8 * k], e[8 * k + 1]), pack2(e[8 * k + 2], e[8 * k + 3]), pack2(e[8 * k + 4], e[8 * k + 5]), pack2(e[8 * k + 6], e[8 * k + 7])};
        }
      }
    }
    if constexpr (SAVE) {                                         // K2: the strip's rows (128 bytes each) as 16-byte stores (the wave's own LDS writes: program order)
#pragma unroll 2
      for (int it = 0; it < 8; ++it) {
        const int idx = lane + 64 * it, r = idx >> 3, k = idx & 7, gr = grow(r);
        if (gr >= 0) *reinterpret_cast<u32x4*>(karg<us16>(KOFF(RgFwd2Args, f.KV16)) + (size_t)gr * 512 + 64 * w + 8 * k) = *reinterpret_cast<const u32x4*>(strip + r * PS + 16 * k);
      }
    }
    f32x16 vacc[RT][2];
    {
      f32x16 init[2] = {splat16(0.f), splat16(0.f)};              // (inference: the values' bias is added by the combine; training: below)
      st1.template run<false>(act1, sub1, init, vacc);              // same fragments, operands swapped: lane = feature
    }
    st1.prefetch(W1s(), w1pair(2 * w), lane);
    stamp(a.stamps, 4);
    if constexpr (!FOLD) {                                        // training: the values carry their bias (lane = feature) ...
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float vb = cst[Cfg::C_BV + 32 * (2 * w + t) + l31];
#pragma unroll
        for (int s = 0; s < RT; ++s)
#pragma unroll
          for (int i = 0; i < 16; ++i) vacc[s][t][i] += vb;
      }
    }
    if constexpr (SAVE) {                                         // ... and are saved: [32 rows][32 features] through 2 KB of scratch -> 16-byte stores
      us16* vt = reinterpret_cast<us16*>(scr + 2048);
#pragma unroll
      for (int s = 0; s < RT; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int i = 0; i < 16; ++i) vt[acc_row(i, h) * 32 + l31] = f2bf(vacc[s][t][i]);
#pragma unroll
          for (int r2 = 0; r2 < 2; ++r2) {
            const int idx = lane + 64 * r2, row = idx >> 2, ch = idx & 3;
            const u32x4 v = *reinterpret_cast<const u32x4*>(vt + 8 * idx);
            if (row < sub[s].nr) *reinterpret_cast<u32x4*>(karg<us16>(KOFF(RgFwd2Args, f.KV16)) + ((size_t)sub[s].row0 + row) * 512 + 256 + 32 * (2 * w + t) + 8 * ch) = v;
          }
        }
    }
    // one partial per run of sub-tiles of one sample: {max (log2 units), sum, Z[query][feature] = E^T . V2 as bf16} at the run's first tile
    auto store_part = [&](int tile, int t, const f32x16& Z, float L, float m) {
      L += __shfl_xor(L, 32, 64);
      float* part = LATEP(SAVE, float, KOFF(RgFwd2Args, b.part), a.part) + ((size_t)tile * 8 + (2 * w + t)) * (SAVE ? PART_FLOATS_F32 : PART_FLOATS);
      if (lane < 16) { part[lane] = m * LOG2E; part[16 + lane] = L; }
      // rows j < Nk leave as 16-byte stores through the wave's scratch (its own LDS writes: program order)
      if constexpr (SAVE) {
        float* zt = reinterpret_cast<float*>(scr + 2048 * t);          // [16 queries][32 features] fp32
#pragma unroll
        for (int i = 0; i < 8; ++i) zt[acc_row(i, h) * 32 + l31] = Z[i];
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const int idx = lane + 64 * r2;                             // row idx >> 3, chunk idx & 7
          if ((idx >> 3) < Nk) *reinterpret_cast<f32x4*>(part + 32 + 4 * idx) = *reinterpret_cast<const f32x4*>(zt + 4 * idx);
        }
      } else {
        us16* zt = reinterpret_cast<us16*>(scr + 1024 * t);          // [16 queries][32 features] bf16: lane -> row lane >> 2, chunk lane & 3
#pragma unroll
        for (int i = 0; i < 8; ++i) zt[acc_row(i, h) * 32 + l31] = f2bf(Z[i]);
        if ((lane >> 2) < Nk) *reinterpret_cast<u32x4*>(reinterpret_cast<us16*>(part + 32) + 8 * lane) = *reinterpret_cast<const u32x4*>(zt + 8 * lane);
      }
    };
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ef[0][t][0]), as_frag(pack8(vacc[0][t], 0)), splat16(0.f), 0, 0, 0);
      Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ef[0][t][1]), as_frag(pack8(vacc[0][t], 1)), Z, 0, 0, 0);
      if (same) {                                                 // (wave-uniform) the common case: both sub-tiles, one partial
        Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ef[1][t][0]), as_frag(pack8(vacc[1][t], 0)), Z, 0, 0, 0);
        Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ef[1][t][1]), as_frag(pack8(vacc[1][t], 1)), Z, 0, 0, 0);
        store_part(g0, t, Z, Lsub[0][t] + Lsub[1][t], mseg[0][t]);
      } else {
        store_part(g0, t, Z, Lsub[0][t], mseg[0][t]);
        if (sub[1].nr > 0) {
          f32x16 Z1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ef[1][t][0]), as_frag(pack8(vacc[1][t], 0)), splat16(0.f), 0, 0, 0);
          Z1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(ef[1][t][1]), as_frag(pack8(vacc[1][t], 1)), Z1, 0, 0, 0);
          store_part(g0 + 1, t, Z1, Lsub[1][t], mseg[1][t]);
        }
      }
    }
  }
  // ---- pass q (lane = row) -> RG->KG attention straight from the accumulators -> the wave's strip
  stamp(a.stamps, 5);
  Stage<RT, 2, 16, 2, DEPTH> sto;
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  {
    // the samples' KG keys in accumulator k order (A operand: lane = key) and value rows [16][32 features of the head] (L2 hits by now)
    u32x4 kf[RT][2][2], vkg[RT][2];
#pragma unroll
    for (int s = 0; s < RT; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int jk = l31 & 15;
        const us16* kp = (LATEP(SAVE, const us16, KOFF(RgFwd2Args, b.KV16), a.KV16) + ((size_t)sub[s].b * Nk * 512 + 32 * (2 * w + t))) + ((unsigned)min(jk, Nk - 1) * 512u + 4u * (unsigned)h);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const u32x2 lo = *reinterpret_cast<const u32x2*>(kp + 16 * kk), hi = *reinterpret_cast<const u32x2*>(kp + 16 * kk + 8);
          kf[s][t][kk] = jk < Nk ? u32x4{lo.x, lo.y, hi.x, hi.y} : u32x4{0u, 0u, 0u, 0u};
        }
        const int j = lane >> 2, c = lane & 3;
        vkg[s][t] = u32x4{0u, 0u, 0u, 0u};
        if (j < Nk) vkg[s][t] = *reinterpret_cast<const u32x4*>((LATEP(SAVE, const us16, KOFF(RgFwd2Args, b.KV16), a.KV16) + ((size_t)sub[s].b * Nk * 512 + 256 + 32 * (2 * w + t))) + ((unsigned)j * 512u + 8u * (unsigned)c));
      }
    f32x16 acc[RT][2];
    {
      f32x16 init[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) init[t] = feature_vec(cst + Cfg::C_BQ + 32 * (2 * w + t), h);
      st1.template run<true>(act1, sub1, init, acc);
    }
    sto.prefetch(LATEP(SAVE, const us16, KOFF(RgFwd2Args, b.s[0].Wo), S.Wo), w * (16 * 2), lane);
    stamp(a.stamps, 6);
    // invalid keys (Nk .. 15) leave the softmax through the score product's C operand
    f32x16 kmask;
#pragma unroll
    for (int i = 0; i < 16; ++i) kmask[i] = (i < 8 && acc_row(i, h) >= Nk) ? -1e30f : 0.f;
#pragma unroll
    for (int s = 0; s < RT; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if constexpr (!FOLD) {                                    // training: the queries are scaled BEFORE they are rounded (and saved so)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[s][t][i] *= g.qscale;
        }
        const u32x4 q0 = pack8(acc[s][t], 0), q1 = pack8(acc[s][t], 1);
        if constexpr (SAVE) {                                     // Q16 (pre-scaled) -> 2 KB of the wave's scratch [32 rows][32 features] -> 16-byte stores
          char* qs = scr + 4096 + 2048 * ((2 * s + t) & 1);          // (two slots, alternating: beyond the value scratch's 4 KB)
          char* qd = qs + l31 * 64 + 2 * (4 * h);
          *reinterpret_cast<u32x2*>(qd) = u32x2{q0.x, q0.y}; *reinterpret_cast<u32x2*>(qd + 16) = u32x2{q0.z, q0.w};
          *reinterpret_cast<u32x2*>(qd + 32) = u32x2{q1.x, q1.y}; *reinterpret_cast<u32x2*>(qd + 48) = u32x2{q1.z, q1.w};
#pragma unroll
          for (int r2 = 0; r2 < 2; ++r2) {
            const int idx = lane + 64 * r2, row = idx >> 2, ch = idx & 3;
            if (row < sub[s].nr) *reinterpret_cast<u32x4*>(karg<us16>(KOFF(RgFwd2Args, f.Q16)) + ((size_t)sub[s].row0 + row) * 256 + 32 * (2 * w + t) + 8 * ch) = *reinterpret_cast<const u32x4*>(qs + 64 * row + 16 * ch);
          }
        }
        f32x16 Sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(kf[s][t][0]), as_frag(q0), kmask, 0, 0, 0);     // S^T[key][row]: lane = row
        Sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(kf[s][t][1]), as_frag(q1), Sc, 0, 0, 0);
        float m = fmaxf(fmaxf(fmaxf(Sc[0], Sc[1]), fmaxf(Sc[2], Sc[3])), fmaxf(fmaxf(Sc[4], Sc[5]), fmaxf(Sc[6], Sc[7])));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float nm = -m * sc2;
        float e[8], sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { e[i] = fexp2(fmaf(Sc[i], sc2, nm)); sum += e[i]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = frcp(sum);
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] *= inv;
        if constexpr (DROP) {
          const uint32_t ibase = ((uint32_t)(sub[s].row0 + l31) * 8u + (uint32_t)(2 * w + t)) * (uint32_t)Nk;
#pragma unroll
          for (int i = 0; i < 8; ++i) e[i] *= drop_mult(a.drop, SITE_ATTN_RG2KG, ibase + (uint32_t)acc_row(i, h));
        }
        const bf16x8 pf = as_frag(u32x4{pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])});
        char* vs = scr + 1024 * (2 * s + t);                      // 1 KB scratch for the transposing read: [16 keys][64 bytes], linear
        *reinterpret_cast<u32x4*>(vs + 16 * lane) = vkg[s][t];
        const char* vp = vs + (4 * h + q4) * 64 + 2 * (16 * g1 + 4 * p4);
        const bf16x8 vf = join(lds_tr16(vp), lds_tr16(vp + 8 * 64));
        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, splat16(0.f), 0, 0, 0);       // O^T[feature][row]
      }
    if constexpr (FOLD) __syncthreads();                          // every wave is done with the input tile (the folded passes' operand): the strips may be written
#pragma unroll
    for (int s = 0; s < RT; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *reinterpret_cast<u32x2*>(strip + (32 * s + l31) * PS + 2 * (32 * t + 8 * gq + 4 * h)) =
              u32x2{pack2(acc[s][t][4 * gq], acc[s][t][4 * gq + 1]), pack2(acc[s][t][4 * gq + 2], acc[s][t][4 * gq + 3])};
  }
  stamp(a.stamps, 7);
  __syncthreads();                                                // strips complete
  stamp(a.stamps, 8);
  if constexpr (SAVE) {                                           // O16: this wave's strip [64 rows][64 features] as 16-byte stores (the wave's own LDS writes: program order)
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
      const int idx = lane + 64 * it, r = idx >> 3, k = idx & 7, gr = grow(r);
      if (gr >= 0) *reinterpret_cast<u32x4*>(karg<us16>(KOFF(RgFwd2Args, b.s[0].O16)) + (size_t)gr * 256 + 64 * w + 8 * k) = *reinterpret_cast<const u32x4*>(strip + r * PS + 16 * k);
    }
  }
  // attention output fragment of sub-tile s, k step ks (features 16 ks ..): strip ks >> 2, byte 32 (ks & 3) of the row
  chain<RT, DEPTH, false, DROP, SAVE>(S, KOFF(RgFwd2Args, b.s[0]), sub, [&](int s, int ks) { return *reinterpret_cast<const bf16x8*>(strips + (ks >> 2) * Cfg::STRIP + (32 * s + l31) * PS + 32 * (ks & 3) + 16 * h); },
                                      grow, a.drop, bufR, strips, red, cst, w, lane, sto, a.stamps);
  stamp(a.stamps, 12);
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int DEPTH, bool FOLD, bool DROP, bool SAVE>
void wide2_launch(const RgFwd2Args& g, dim3 grid, hipStream_t stream) {
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rgfwd2_kernel<DEPTH, FOLD, DROP, SAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    return true;
  }();
  (void)attr;
  hipLaunchKernelGGL((rgfwd2_kernel<DEPTH, FOLD, DROP, SAVE>), grid, dim3(NTH), Cfg::LDS, stream, g);
}
template <int DEPTH, bool DROP, bool SAVE>
void kgchain_launch(const KgChainArgs& k, hipStream_t stream) {
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kgchain_kernel<DEPTH, DROP, SAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    return true;
  }();
  (void)attr;
  hipLaunchKernelGGL((kgchain_kernel<DEPTH, DROP, SAVE>), dim3((k.b.B + 1) / 2), dim3(NTH), Cfg::LDS, stream, k);
}

}  // namespace

int wide2_max_rows() { return 4096; }      // (the fused schedule's own limit: nothing here depends on a sample's length)

// The RG rows' whole forward on 64-row half-blocks and, behind it, the KG rows' launch (two samples per block).  Inference calls
// (b.save == 0, no dropout) take the folded in-projection Wf / bf; training calls (b.save, dropout) the unfolded shadow f.W1 and
// write the backward's saved set.  `f` = the RG stream of the front half, `b` = the back half's arguments.  The KG rows' projections
// (b.Q2_16, b.KV16, b.s[1].R16) must already exist: launch_wide_front(..., kg_only = 1) first.
// Wf / bf of the folded in-projection (see fold_rg_kernel): Wq [256 x 256], Wkv [512 x 256] = [Wk2; Wv2], their biases, the projection
// Wrg [256 x 128] / brg -> Wf (bf16 shadow of [768 x 128], 196 608 bytes), bf [768].
int launch_fold_rg(const float* Wq, const float* Wkv, const float* bq, const float* bkv, const float* Wrg, const float* brg, us16* Wf, float* bf, hipStream_t stream) {
  if (!Wq || !Wkv || !bq || !bkv || !Wrg || !brg || !Wf || !bf || !al16(Wq) || !al16(Wkv) || !al16(Wrg) || !al16(Wf)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(fold_rg_kernel, dim3((768 * 128 / 8 + 768 + 255) / 256), dim3(256), 0, stream, Wq, Wkv, bq, bkv, Wrg, brg, Wf, bf);
  return (int)hipGetLastError();
}

int launch_wide2_rgfwd(const FrontStream& f, const us16* Wf, const float* bf, float qscale, BackArgs& b, int max_nr, int save_r16, hipStream_t stream) {
  if (b.B < 1 || b.Nk < 1 || b.Nk > 16 || b.rg_tiles_max < 1 || !b.KV16 || !b.Q2_16 || !b.off || !b.tile_off || !b.tile_desc || !b.inv_nr || !b.part)
    return (int)hipErrorInvalidValue;
  if (max_nr > wide2_max_rows()) return (int)hipErrorInvalidValue;
  const bool save = b.save != 0, drop = b.drop.p > 0.f, fold = !save && !drop;      // (the folded in-projection serves plain inference calls)
  if (!f.X || !f.W0 || !f.b0 || !al16(f.X) || !al16(f.b0) || !al16(f.W0)) return (int)hipErrorInvalidValue;
  if (fold && (!Wf || !bf || !al16(Wf) || !al16(bf))) return (int)hipErrorInvalidValue;
  if (!fold && (!f.W1 || !f.bq || !f.bkv || !al16(f.W1) || !al16(f.bq) || !al16(f.bkv))) return (int)hipErrorInvalidValue;
  if (save && (!f.X16 || !f.R16 || !f.Q16 || !f.KV16 || !al16(f.X16) || !al16(f.R16) || !al16(f.Q16) || !al16(f.KV16))) return (int)hipErrorInvalidValue;
  for (int i = 0; i < 2; ++i) {
    const BackStream& S = b.s[i];
    if (!S.Wo || !S.bo || !S.W1 || !S.b1 || !S.ln_g || !S.ln_b || !S.Ymean || !S.Hmean || (i == 1 && !S.R16)) return (int)hipErrorInvalidValue;
    if (!al16(S.bo) || !al16(S.ln_g) || !al16(S.ln_b) || !al16(S.b1) || !al16(S.Wo) || !al16(S.W1) || (i == 1 && !al16(S.R16))) return (int)hipErrorInvalidValue;
    if (save && (!S.O16 || !S.Y16 || !S.XH16 || !S.rstd || !S.mask || !al16(S.O16) || !al16(S.Y16) || !al16(S.XH16))) return (int)hipErrorInvalidValue;
  }
  RgFwd2Args g; g.f = f; g.b = b; g.qscale = qscale; g.Wf = Wf; g.bf = bf; g.save_r16 = save_r16;
  KgChainArgs k; k.b = b; k.bv2 = fold ? bf + 512 : nullptr;      // (unfolded: the values carry their bias)
  k.b.stamps = b.stamps ? b.stamps + (size_t)(b.rg_tiles_max + RT - 1) / RT * 8 * 16 : nullptr;      // (timeline: the KG blocks' rows follow the RG blocks')
  // executed FLOPs per RG row: 128 -> 256, 128 | 256 -> 768, 256 -> 256, 256 -> 512 and both attention directions
  const double rows = (double)b.rows_rg, kgrows = (double)b.B * b.Nk;
  const dim3 grid((b.rg_tiles_max + RT - 1) / RT);
  int prof = gemm_prof_open(stream, 2.0 * rows * (128.0 * 256.0 + (fold ? 128.0 : 256.0) * 768.0 + 256.0 * 256.0 + 256.0 * 512.0) + 8.0 * rows * b.Nk * 256.0, PROF_BACK);
  if (fold) {
    // weight fragments in flight per wave (b.exp: developer A/B of the prefetch depth; product calls pass 0)
    if (b.exp == 8) wide2_launch<8, true, false, false>(g, grid, stream); else if (b.exp == 16) wide2_launch<16, true, false, false>(g, grid, stream);
    else wide2_launch<12, true, false, false>(g, grid, stream);
  } else if (save) {
    if (b.exp == 8) { if (drop) wide2_launch<8, false, true, true>(g, grid, stream); else wide2_launch<8, false, false, true>(g, grid, stream); }
    else if (drop) wide2_launch<6, false, true, true>(g, grid, stream); else wide2_launch<6, false, false, true>(g, grid, stream);
  } else {
    wide2_launch<8, false, true, false>(g, grid, stream);         // (an inference call in training mode: dropout, nothing saved)
  }
  gemm_prof_close(prof, stream);
  int e = (int)hipGetLastError();
  if (e) return e;
  prof = gemm_prof_open(stream, 2.0 * kgrows * (256.0 * 256.0 + 256.0 * 512.0), PROF_BACK);
  if (save) { if (drop) kgchain_launch<8, true, true>(k, stream); else kgchain_launch<8, false, true>(k, stream); }
  else if (drop) kgchain_launch<8, true, false>(k, stream);
  else kgchain_launch<12, false, false>(k, stream);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}
