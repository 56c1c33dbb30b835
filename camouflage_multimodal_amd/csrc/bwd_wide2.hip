// Backward, first half, for the RG rows of large batches (gfx950): 64-row half-blocks of 4 waves, two independent blocks per CU -- the
// layout of fused_wide2.hip (forward) applied to bwd1_kernel of fused_rows.hip, whose contract (Bwd1Args, fused_rows.h), weight shadows
// (W1T, WoT), saved set and outputs it keeps bit-compatible:
//   dH = mask(H) * d(mean H) / n (* dropout scale)      -> dH16 (the weight-gradient operand), and the first product's activation
//   dY = d(mean Z) / n + dH . W1 ;  dU = LayerNorm_backward(dY) (+ dgamma, dbeta) -> dU16 ;  dO = dU . Wo
//   RG->KG attention backward (probabilities recomputed from the saved queries): dQ -> dQKV16[:, 0..255], dK | dV += (fp32 atomics)
// The KG rows, and the dH16 rows of the KG stream, stay with bwd1_kernel (launched for them alone: launch_fused_bwd1's rg_tiles = 0 form).
//
// Why.  bwd1_kernel is the largest kernel of a large-batch training step (32 % at B = 1024) at ~10 % MFMA issue: a
// 32-row tile streams 384 KB of weights, every one of its 8 waves expands the same 512 mask bits per row into dH fragments (24 vector
// instructions per fragment: 6 per MFMA) and separate writer blocks expand them once more for dH16.  Here
//   * a block owns two 32-row sub-tiles: a weight fragment feeds two MFMAs (half the L2 weight traffic per row), and two blocks per CU
//     drift through their phases independently;
//   * the dH tile [64][512] (bf16) is built ONCE per block in LDS, cooperatively (4 vector instructions per feature pair, one pass),
//     read by every wave as an ordinary activation tile, and written to dH16 from LDS as whole-row 16-byte stores: no writer blocks
//     for the RG rows, no second read of the mask;
//   * d(mean Z) / n enters as the first MFMA's C operand;
//   * the LayerNorm gradient's column sums (dgamma, dbeta) are MFMAs against an identity fragment on hi / lo bf16 planes (exact to
//     2^-17) of the wave's own 64 features, instead of an fp32 tile in LDS summed by one thread per column;
//   * the attention backward runs per sub-tile out of per-wave strips (queries, dO: the wave's 64 features only).
#include "fused_rows.h"
#include "gemm.h"      // launch timing hooks
#include "wide2_inl.h"

namespace {

constexpr int PH = 1040;     // row pitch (bytes) of the [rows][512] bf16 dH tile (65 x 16 bytes: 8 consecutive rows on 8 distinct 16-byte slots)

// NT = feature tiles (= attention heads) per wave: 2 -> blocks of 4 waves (the forward's shape), 1 -> blocks of 8 waves on the same 64
// rows and the same LDS: four waves per SIMD instead of two.  The kernel is latency-bound at two (PMC at B = 1024: vector issue 15 %,
// MFMA 9 % of a wave's cycles, 46 % waiting for an instruction to issue), so the product build takes NT = 1.
template <int NT>
struct CfgB {
  static constexpr int NWB = 8 / NT, NTHB = 64 * NWB;
  static constexpr int PSW = 64 * NT + 16;                       // row pitch (bytes) of a per-wave strip [rows][32 NT features]
  static constexpr int GT = 0;                                   // [RT][512] bf16: d(mean H) / n * dropout scale of each sub-tile's sample
  static constexpr int RED = 0;                                  // LayerNorm gradient partials {sum g, sum g xhat} per (wave, row): over the table (dead by then)
  static constexpr int CST = 4096;                               // floats: ln_g [256] | d(mean Z) / n of sub-tile 0 [256] | of sub-tile 1 [256]
  static constexpr int DU = CST + 3 * 1024;                      // dU tile [ROWS][256] bf16 (pitch PR); then per-wave attention scratch (WSCR each)
  static constexpr int ST = DU + ROWS * PR;                      // per-wave strips [ROWS][32 NT] (pitch PSW): column-sum planes; then queries | dO
  static constexpr int STRIP = ROWS * PSW;
  static constexpr int LDS = ST + NWB * STRIP;
  static constexpr int WSCR = ROWS * PR / NWB;                   // keys [16][32 NT] (pitch PSW) | dS, Pd images [32][16] of each of the wave's heads
  static constexpr int W_KS = 0, W_IMG = 16 * PSW;
  static_assert(RT * 1024 <= 4096 && NWB * ROWS * 8 <= 4096, "table / partials in front of the constants");
  static_assert(ROWS * PH <= ROWS * PR + NWB * STRIP, "the dH tile aliases [dU tile | strips]");
  static_assert(W_IMG + NT * 2048 <= WSCR, "per-wave attention scratch");
  static_assert(LDS <= 81920, "two blocks per CU");
};

// Softmax over the <= 16 keys of one RG row -- the code of fused_rows.hip (rg_softmax), which the saved probabilities' consumers share
__device__ __forceinline__ void rg_softmax(const f32x16& Sc, int h, int Nk, float (&p)[8]) {
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 8; ++i) { p[i] = acc_row(i, h) < Nk ? Sc[i] : -INFINITY; m = fmaxf(m, p[i]); }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { p[i] = __expf(p[i] - m); sum += p[i]; }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] *= inv;
}

template <int DEPTH, bool DROP, int NT>
__global__ __launch_bounds__(CfgB<NT>::NTHB, CfgB<NT>::NWB / 2) void bwd1w_kernel(const Bwd1Args a) {      // (HIP: threads per block, min WAVES PER SIMD -- two blocks per CU)
  using C = CfgB<NT>;
  constexpr int NWB = C::NWB, NTHB = C::NTHB, PSW = C::PSW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Bwd1Stream& S = a.s[0];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, Nk = a.Nk;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ft0 = NT * w;                                         // this wave's first 32-feature tile (= its first head) of the 256-wide layers
  us16* gtab = reinterpret_cast<us16*>(smem + C::GT);
  float* cst = reinterpret_cast<float*>(smem + C::CST);
  float* red = reinterpret_cast<float*>(smem + C::RED);
  char* tileH = smem + C::DU;                                     // dH tile (first product), aliasing [dU tile | strips]
  char* bufdU = smem + C::DU;
  char* strip = smem + C::ST + w * C::STRIP;
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
  // weight fragments: the shadows' 4-wave layout -- wave w4 = tile >> 1 owns fragments (k step, tile & 1) at w4 (2 KS) + 2 ks + (tile & 1)
  Stage<RT, NT, 32, 2, DEPTH> st1;
  st1.prefetch(S.W1T, (ft0 >> 1) * (32 * 2) + (ft0 & 1), lane);
  // ---- per-sample tables: the FFN gradient row as bf16 (what a set mask bit selects), d(mean Z) / n, gamma
  {
    const float gsc = a.drop.scale;
#pragma unroll
    for (int it = 0; it < RT * 512 / NTHB; ++it) {
      const int s = (NTHB * it) >> 9, f = (tid + NTHB * it) & 511;
      gtab[512 * s + f] = f2bf(S.dHm[(size_t)sub[s].b * S.ld_dHm + f] * (sub[s].inv_n * gsc));
    }
    if (tid < 256) {
      cst[tid] = S.ln_g[tid];
#pragma unroll
      for (int s = 0; s < RT; ++s) cst[256 + 256 * s + tid] = S.dcomb[(size_t)sub[s].b * S.ld_dcomb + tid] * sub[s].inv_n;
    }
  }
  // this thread's share of the mask: WPT words of one row (a wave's threads cover rows of ONE sub-tile)
  constexpr int WPT = 1024 / NTHB, TPR = 16 / WPT;                // words per thread (4 or 2), threads per row
  const bool hs1 = w * (64 / TPR) >= 32;                          // (wave-uniform) this wave's rows are sub-tile 1's
  const int hs_row0 = hs1 ? sub[1].row0 : sub[0].row0, hs_nr = hs1 ? sub[1].nr : sub[0].nr;
  const int hr = tid / TPR, hq = tid % TPR, hrr = hr & 31;
  uint32_t mw[WPT];
#pragma unroll
  for (int j = 0; j < WPT; ++j) mw[j] = 0u;
  if (hrr < hs_nr) {
    const uint32_t* mp = S.mask + ((size_t)hs_row0 + hrr) * 16 + WPT * hq;
    if constexpr (WPT == 4) { const u32x4 v = *reinterpret_cast<const u32x4*>(mp); mw[0] = v.x; mw[1] = v.y; mw[2] = v.z; mw[3] = v.w; }
    else                    { const u32x2 v = *reinterpret_cast<const u32x2*>(mp); mw[0] = v.x; mw[1] = v.y; }
  }
  // this lane's rows of the two sub-tiles (clamped into the tile), their saved normalised LayerNorm inputs and 1 / std
  // The normalised inputs arrive as 16-byte chunks, a wave's 32 NT features of 64 rows in 4 NT coalesced instructions (one 8-byte piece
  // per lane, row and feature group touches 32 cache lines per instruction: the kernel is bound by the vector memory pipeline -- PMC at
  // B = 1024: TA busy 76 % -- and those loads were a sixth of its line traffic); they pass through the wave's own strip once the dH
  // tile is dead and come back in the accumulators' layout (lane = row).
  constexpr int CPR = 4 * NT;                                     // 16-byte chunks of this wave's features per row
  bool rok[RT]; u32x2 xv[RT][NT][4]; float rstd[RT]; u32x4 xch[RT][2 * NT];
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    rok[s] = l31 < sub[s].nr;
    const size_t vrow = (size_t)sub[s].row0 + max(0, min(l31, sub[s].nr - 1));
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) {
      const int c = lane + 64 * i;
      xch[s][i] = *reinterpret_cast<const u32x4*>(S.XH16 + ((size_t)sub[s].row0 + max(0, min(c / CPR, sub[s].nr - 1))) * 256 + 32 * ft0 + 8 * (c % CPR));
    }
    rstd[s] = S.rstd[vrow];
  }
  __syncthreads();                                                // tables complete
  // ---- dH tile: feature pair (2 d, 2 d + 1) of a 16-byte chunk <- the table's pair where the mask bits are set
  {
    const char* gt = reinterpret_cast<const char*>(gtab) + (hs1 ? 1024 : 0) + 64 * WPT * hq;
    char* dst = tileH + hr * PH + 64 * WPT * hq;
#pragma unroll
    for (int j = 0; j < WPT; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t bits = mw[j] >> (8 * i);
        const u32x4 gv = *reinterpret_cast<const u32x4*>(gt + 64 * j + 16 * i);
        u32x4 fr;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const uint32_t lo = (uint32_t)((int)(bits << (31 - 2 * d)) >> 31), hi = (uint32_t)((int)(bits << (30 - 2 * d)) >> 31);
          fr[d] = gv[d] & ((lo & 0xFFFFu) | (hi & 0xFFFF0000u));
        }
        *reinterpret_cast<u32x4*>(dst + 64 * j + 16 * i) = fr;
      }
  }
  __syncthreads();                                                // dH tile complete
  stamp(a.stamps, 1);
  // dH16: whole rows as 16-byte stores (a wave writes one 1 KB row per instruction)
#pragma unroll 4
  for (int it = 0; it < ROWS / NWB; ++it) {
    const int r = w + NWB * it, rr = r & 31;
    const bool s1 = it >= 32 / NWB;                                // (scalar selects: no indexed sub[])
    const int nr_s = s1 ? sub[1].nr : sub[0].nr, row0_s = s1 ? sub[1].row0 : sub[0].row0;
    if (rr < nr_s) *reinterpret_cast<u32x4*>(S.dH16 + ((size_t)row0_s + rr) * 512 + 8 * lane) = *reinterpret_cast<const u32x4*>(tileH + r * PH + 16 * lane);
  }
  // ---- dY = d(mean Z) / n + dH . W1 (lane = row, registers = features 32 (ft0 + t) + acc_row)
  f32x16 acc1[RT][NT];
  st1.template run_fi<true>([&](int s, int ks) { return *reinterpret_cast<const bf16x8*>(tileH + (32 * s + l31) * PH + 16 * h + 32 * ks); },
                            [&](int s, int t) { return feature_vec(cst + 256 + 256 * s + 32 * (ft0 + t), h); }, acc1);
  stamp(a.stamps, 2);
  Stage<RT, NT, 16, 2, DEPTH> st2;
  __syncthreads();                                                // every wave is done with the dH tile (the strips lie inside it)
#pragma unroll
  for (int s = 0; s < RT; ++s)
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) { const int c = lane + 64 * i; *reinterpret_cast<u32x4*>(strip + (32 * s + c / CPR) * PSW + 16 * (c % CPR)) = xch[s][i]; }
#pragma unroll
  for (int s = 0; s < RT; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) xv[s][t][g] = *reinterpret_cast<const u32x2*>(strip + (32 * s + l31) * PSW + 2 * (32 * t + 8 * g + 4 * h));
  // ---- LayerNorm backward: g = dy gamma; du = (g - mean(g) - xhat mean(g xhat)) / std; dgamma += dy xhat, dbeta += dy
  // rows past a sub-tile's end: dy and xhat cleared once (then g, both row sums and du are exactly 0 there and the column sums skip
  // them); full sub-tiles -- the common case -- carry no masks at all
  if (sub[0].nr < 32 || sub[1].nr < 32) {                           // (wave-uniform)
#pragma unroll
    for (int s = 0; s < RT; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[s][t][i] = rok[s] ? acc1[s][t][i] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) xv[s][t][g] = rok[s] ? xv[s][t][g] : u32x2{0u, 0u};
      }
  }
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(cst + 32 * (ft0 + t) + 8 * g + 4 * h);
        const u32x2 x = xv[s][t][g];
        const float X[4] = {bf_lo(x.x), bf_hi(x.x), bf_lo(x.y), bf_hi(x.y)};
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float G = acc1[s][t][4 * g + j] * gm[j]; s1 += G; s2 = fmaf(G, X[j], s2); }
      }
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    if (h == 0) *reinterpret_cast<float2*>(red + 2 * (w * ROWS + 32 * s + l31)) = make_float2(s1, s2);
  }
  __syncthreads();                                                // partials complete
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWB; ++ww) {
      const float2 p = *reinterpret_cast<const float2*>(red + 2 * (ww * ROWS + 32 * s + l31));
      m1 += p.x; m2 += p.y;
    }
    const float nm1 = -m1 * (1.0f / 256.0f), nm2 = -m2 * (1.0f / 256.0f);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 32 * (ft0 + t) + 8 * g + 4 * h;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(cst + c0);
        const u32x2 x = xv[s][t][g];
        const float X[4] = {bf_lo(x.x), bf_hi(x.x), bf_lo(x.y), bf_hi(x.y)};
        float du[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) du[j] = fmaf(X[j], nm2, fmaf(acc1[s][t][4 * g + j], gm[j], nm1)) * rstd[s];
        *reinterpret_cast<u32x2*>(bufdU + (32 * s + l31) * PR + 2 * c0) = u32x2{pack2(du[0], du[1]), pack2(du[2], du[3])};
      }
  }
  // column sums over the block's rows of dy xhat (dgamma) and dy (dbeta), this wave's features.  The two sub-tiles' values of one
  // (row slot, feature) are added in the lane first; the sums go through the wave's strip as bf16 hi planes (dy xhat: strip rows
  // 0 .. 31, dy: rows 32 .. 63), then lo planes (exact to 2^-17), and come back as MFMAs against the identity: lane = feature,
  // registers = row slots -- an fp32 tile in LDS summed by one thread per column costs 64 KB for these rows
  {
    u32x4 idf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int j = l31 - 16 * kk - 8 * h;
      const uint32_t one_lo = 0x3F80u, one_hi = 0x3F800000u;
      idf[kk] = u32x4{j == 0 ? one_lo : (j == 1 ? one_hi : 0u), j == 2 ? one_lo : (j == 3 ? one_hi : 0u),
                      j == 4 ? one_lo : (j == 5 ? one_hi : 0u), j == 6 ? one_lo : (j == 7 ? one_hi : 0u)};
    }
    u32x2 lo[2][NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u32x2 x0 = xv[0][t][g], x1 = xv[1][t][g];
        const float X0[4] = {bf_lo(x0.x), bf_hi(x0.x), bf_lo(x0.y), bf_hi(x0.y)}, X1[4] = {bf_lo(x1.x), bf_hi(x1.x), bf_lo(x1.y), bf_hi(x1.y)};
        float v0[4], v1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v0[j] = fmaf(acc1[0][t][4 * g + j], X0[j], acc1[1][t][4 * g + j] * X1[j]);
          v1[j] = acc1[0][t][4 * g + j] + acc1[1][t][4 * g + j];
        }
        const u32x2 h0 = u32x2{pack2(v0[0], v0[1]), pack2(v0[2], v0[3])}, h1 = u32x2{pack2(v1[0], v1[1]), pack2(v1[2], v1[3])};
        lo[0][t][g] = u32x2{pack2(v0[0] - bf_lo(h0.x), v0[1] - bf_hi(h0.x)), pack2(v0[2] - bf_lo(h0.y), v0[3] - bf_hi(h0.y))};
        lo[1][t][g] = u32x2{pack2(v1[0] - bf_lo(h1.x), v1[1] - bf_hi(h1.x)), pack2(v1[2] - bf_lo(h1.y), v1[3] - bf_hi(h1.y))};
        *reinterpret_cast<u32x2*>(strip + l31 * PSW + 2 * (32 * t + 8 * g + 4 * h)) = h0;
        *reinterpret_cast<u32x2*>(strip + (32 + l31) * PSW + 2 * (32 * t + 8 * g + 4 * h)) = h1;
      }
    f32x16 p[2][NT];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      if (pl == 1) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<u32x2*>(strip + (32 * q + l31) * PSW + 2 * (32 * t + 8 * g + 4 * h)) = lo[q][t][g];
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const char* yp = strip + (32 * q + l31) * PSW + 64 * t + 16 * h;
          p[q][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(yp), as_frag(idf[0]), pl == 0 ? splat16(0.f) : p[q][t], 0, 0, 0);
          p[q][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(yp + 32), as_frag(idf[1]), p[q][t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float c = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) c += p[q][t][i];
        c += __shfl_xor(c, 32, 64);
        if (h == 0) atomicAdd((q == 0 ? S.dgamma : S.dbeta) + 32 * (ft0 + t) + l31, c);
      }
  }
  st2.prefetch(S.WoT, (ft0 >> 1) * (16 * 2) + (ft0 & 1), lane);  // (in front of the barrier: the column sums above need the registers)
  // the first sub-tile's attention operands start now and land under the dO product: queries (the wave's 32 NT features of 32 rows,
  // 4 NT chunks of 16 bytes per row), the sample's keys (-> LDS, for the transposing reads) and its values as A fragments straight from L2
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  u32x4 qreg[2 * NT], kreg[NT]; u32x2 vfr[NT][2][2];
  auto load_q = [&](int row0, int nr) {
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) {
      const int c = lane + 64 * i;
      qreg[i] = *reinterpret_cast<const u32x4*>(a.Q16 + ((size_t)row0 + max(0, min(c / CPR, nr - 1))) * 256 + 32 * ft0 + 8 * (c % CPR));
    }
  };
  auto load_kv = [&](int b) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int c = lane + 64 * i, j = c / CPR;
      kreg[i] = *reinterpret_cast<const u32x4*>(a.KV16 + ((size_t)b * Nk + min(j, Nk - 1)) * 512 + 32 * ft0 + 8 * (c % CPR));
      if (j >= Nk) kreg[i] = u32x4{0u, 0u, 0u, 0u};
    }
    // value row j = lane & 15, features 16 ss + 4 h .. + 3 and + 8 .. of head t: the k order of the dO accumulator's registers
    const int j = l31 & 15;
    const us16* vp = a.KV16 + ((size_t)b * Nk + min(j, Nk - 1)) * 512 + 256 + 32 * ft0 + 4 * h;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        vfr[t][ss][0] = *reinterpret_cast<const u32x2*>(vp + 32 * t + 16 * ss);
        vfr[t][ss][1] = *reinterpret_cast<const u32x2*>(vp + 32 * t + 16 * ss + 8);
        if (j >= Nk) { vfr[t][ss][0] = u32x2{0u, 0u}; vfr[t][ss][1] = u32x2{0u, 0u}; }
      }
  };
  if constexpr (NT == 2) { load_q(sub[0].row0, sub[0].nr); load_kv(sub[0].b); }     // (8 waves: behind the dO product -- 128 registers)
  __syncthreads();                                                // dU tile complete
  stamp(a.stamps, 3);
  // ---- dO = dU . Wo (lane = row, registers = the features of this wave's heads)
  f32x16 acc2[RT][NT];
  {
    f32x16 init[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) init[t] = splat16(0.f);
    st2.template run<true>(bufdU + l31 * PR + 16 * h, 32 * PR, init, acc2);
  }
  if constexpr (NT == 1) { load_q(sub[0].row0, sub[0].nr); load_kv(sub[0].b); }
  __syncthreads();                                                // every wave is done reading the dU tile
  stamp(a.stamps, 4);
#pragma unroll 4
  for (int it = 0; it < ROWS * 32 / NTHB; ++it) {                 // dU16: the tile's rows as 16-byte stores (behind the block's last weight stream)
    const int r = (tid >> 5) + (NTHB / 32) * it, rr = r & 31, k = tid & 31;
    const bool s1 = it >= ROWS * 16 / NTHB;
    const int nr_s = s1 ? sub[1].nr : sub[0].nr, row0_s = s1 ? sub[1].row0 : sub[0].row0;
    if (rr < nr_s) store16_wt(S.dU16 + ((size_t)row0_s + rr) * 256 + 8 * k, *reinterpret_cast<const u32x4*>(bufdU + r * PR + 16 * k));
  }
  __syncthreads();                                                // the dU tile's space becomes per-wave scratch
  stamp(a.stamps, 5);
  // ---- RG->KG attention backward, one sub-tile at a time, out of this wave's own LDS (no block barriers from here on); the heads
  // of a wave share nothing but read-only tiles (their own dS / Pd images), so their chains overlap
  char* Qs = strip; char* dOs = strip + 32 * PSW;
  char* wscr = smem + C::DU + w * C::WSCR;
  char* Ks = wscr + C::W_KS;
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    if (sub[s].nr == 0) break;
    const size_t rowg0 = (size_t)sub[s].row0;
    if (s == 0 || sub[s].b != sub[0].b) {                         // (wave-uniform) a new sample's keys
#pragma unroll
      for (int i = 0; i < NT; ++i) { const int c = lane + 64 * i; *reinterpret_cast<u32x4*>(Ks + (c / CPR) * PSW + 16 * (c % CPR)) = kreg[i]; }
    }
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) { const int c = lane + 64 * i; *reinterpret_cast<u32x4*>(Qs + (c / CPR) * PSW + 16 * (c % CPR)) = qreg[i]; }
    u32x2 vf[NT][2][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) { vf[t][ss][0] = vfr[t][ss][0]; vf[t][ss][1] = vfr[t][ss][1]; }
    if (s + 1 < RT && sub[RT - 1].nr > 0) {                       // the next sub-tile's operands, under this one's attention
      load_q(sub[RT - 1].row0, sub[RT - 1].nr);
      if (sub[RT - 1].b != sub[0].b) load_kv(sub[RT - 1].b);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)                                  // dO strip (bf16): read back transposed for dV
        *reinterpret_cast<u32x2*>(dOs + l31 * PSW + 2 * (32 * t + 8 * g + 4 * h)) =
            rok[s] ? u32x2{pack2(acc2[s][t][4 * g], acc2[s][t][4 * g + 1]), pack2(acc2[s][t][4 * g + 2], acc2[s][t][4 * g + 3])} : u32x2{0u, 0u};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int head = ft0 + t;
      char* imS = wscr + C::W_IMG + 2048 * t; char* imP = imS + 1024;
      // scores and probabilities, exactly as bwd1_kernel recomputes them
      f32x16 Sc = splat16(0.f);
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (l31 & 15) * PSW + 2 * (32 * t + 16 * ss + 8 * h));
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qs + l31 * PSW + 2 * (32 * t + 16 * ss + 8 * h));
        Sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf, Sc, 0, 0, 0);
      }
      float pr[8], mm[8];
      rg_softmax(Sc, h, Nk, pr);
      const uint32_t ibase = ((uint32_t)(rowg0 + l31) * 8u + (uint32_t)head) * (uint32_t)Nk;
#pragma unroll
      for (int i = 0; i < 8; ++i) mm[i] = DROP ? drop_mult(a.drop, SITE_ATTN_RG2KG, ibase + (uint32_t)acc_row(i, h)) : 1.0f;
      // dPd^T[j][row] = V_h[j] . dO_h[row]: the dO accumulator is the B operand (k order of its registers)
      f32x16 dP = splat16(0.f);
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
        dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(u32x4{vf[t][ss][0].x, vf[t][ss][0].y, vf[t][ss][1].x, vf[t][ss][1].y}), as_frag(pack8(acc2[s][t], ss)), dP, 0, 0, 0);
      float ds[8], pd[8], delta = 0.f;
      f32x16 dq;
#pragma unroll
      for (int i = 0; i < 8; ++i) { ds[i] = dP[i] * mm[i]; delta = fmaf(pr[i], ds[i], delta); pd[i] = pr[i] * mm[i]; }
      delta += __shfl_xor(delta, 32, 64);
#pragma unroll
      for (int i = 0; i < 8; ++i) { ds[i] = rok[s] ? pr[i] * (ds[i] - delta) : 0.f; if (!rok[s]) pd[i] = 0.f; }
      const u32x4 dsf = u32x4{pack2(ds[0], ds[1]), pack2(ds[2], ds[3]), pack2(ds[4], ds[5]), pack2(ds[6], ds[7])};
      const u32x4 pdf = u32x4{pack2(pd[0], pd[1]), pack2(pd[2], pd[3]), pack2(pd[4], pd[5]), pack2(pd[6], pd[7])};
      {   // dQ^T = scale * K_h^T . dS^T: lane = row
        const char* kp = Ks + (4 * h + q4) * PSW + 2 * (32 * t + 16 * g1 + 4 * p4);
        const bf16x8 kt = join(lds_tr16(kp), lds_tr16(kp + 8 * PSW));
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, as_frag(dsf), splat16(0.f), 0, 0, 0);
      }
      // dS / Pd images [row][16 keys] (bf16): keys 4 h .. + 3 at byte 8 h, keys 8 + 4 h .. at byte 16 + 8 h
      *reinterpret_cast<u32x2*>(imS + l31 * 32 + 8 * h) = u32x2{dsf.x, dsf.y};
      *reinterpret_cast<u32x2*>(imS + l31 * 32 + 16 + 8 * h) = u32x2{dsf.z, dsf.w};
      *reinterpret_cast<u32x2*>(imP + l31 * 32 + 8 * h) = u32x2{pdf.x, pdf.y};
      *reinterpret_cast<u32x2*>(imP + l31 * 32 + 16 + 8 * h) = u32x2{pdf.z, pdf.w};
      // dK_h[j][f] += sum_rows dS[j][row] Qs[row][f];  dV_h[j][f] += sum_rows Pd[j][row] dO[row][f]  (lane = f, registers = j)
      f32x16 dK = splat16(0.f), dV = splat16(0.f);
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const int r0 = 16 * ss + 8 * h + q4;
        const bf16x8 sA = join(lds_tr16(imS + r0 * 32 + 8 * p4), lds_tr16(imS + (r0 + 4) * 32 + 8 * p4));
        const bf16x8 pA = join(lds_tr16(imP + r0 * 32 + 8 * p4), lds_tr16(imP + (r0 + 4) * 32 + 8 * p4));
        const int co = 2 * (32 * t + 16 * g1 + 4 * p4);
        const bf16x8 qB = join(lds_tr16(Qs + r0 * PSW + co), lds_tr16(Qs + (r0 + 4) * PSW + co));
        const bf16x8 oB = join(lds_tr16(dOs + r0 * PSW + co), lds_tr16(dOs + (r0 + 4) * PSW + co));
        dK = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sA, qB, dK, 0, 0, 0);
        dV = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pA, oB, dV, 0, 0, 0);
      }
      // dQ rows (bf16, 64 bytes per row and head) leave through the head's image space, dead now, as 16-byte pieces: four lanes per row
      // -- 8-byte pieces straight from the accumulators touch 64 cache lines per store instruction
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<u32x2*>(imS + l31 * 64 + 2 * (8 * g + 4 * h)) =
            u32x2{pack2(dq[4 * g] * a.qscale, dq[4 * g + 1] * a.qscale), pack2(dq[4 * g + 2] * a.qscale, dq[4 * g + 3] * a.qscale)};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = lane + 64 * i, r = c >> 2;
        if (r < sub[s].nr) *reinterpret_cast<u32x4*>(a.dQKV16 + (rowg0 + r) * 768 + 32 * head + 8 * (c & 3)) = *reinterpret_cast<const u32x4*>(imS + r * 64 + 16 * (c & 3));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = acc_row(i, h);
        if (j < Nk) {
          float* dst = a.dKV + ((size_t)sub[s].b * Nk + j) * 512 + 32 * head + l31;
          atomicAdd(dst, dK[i]);
          atomicAdd(dst + 256, dV[i]);
        }
      }
    }
  }
  stamp(a.stamps, 12);
}

template <int DEPTH, bool DROP, int NT>
int bwd1w_launch(const Bwd1Args& a, hipStream_t stream) {
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd1w_kernel<DEPTH, DROP, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, CfgB<NT>::LDS);
    return true;
  }();
  (void)attr;
  hipLaunchKernelGGL((bwd1w_kernel<DEPTH, DROP, NT>), dim3((a.rg_tiles_max + RT - 1) / RT), dim3(CfgB<NT>::NTHB), CfgB<NT>::LDS, stream, a);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ backward, second half (parameter-space form)
// bwd2p_kernel of fused_rows.hip for the RG rows of large batches: the KG->RG attention backward of a block's 64 rows -- probabilities
// recomputed from the saved softmax {max, sum}; dK2 | dV2 per row into columns 256 .. 767 of dQKV16, the sample's dQ2 by fp32 atomics.
// There is no weight stream here, only rows in and rows out (2 KB per row): a wave is one head and keeps everything of its head in its
// own 8.7 KB of LDS -- key and value slices [32 rows][32 features] loaded as 16-byte pieces (4 lanes per row), the sample's 16
// queries and d(attention output) rows, the dS image -- so the block has NO barrier, and the next sub-tile's slices are in flight
// while this one computes.  The dK2 / dV2 rows leave through the dead key / value slices as 16-byte pieces.  (bwd2p_kernel: 8 waves
// on 32 rows between three barriers, a 49 KB tile for 1 KB rows, 73 % of its wave cycles waiting.)
struct CfgC {
  static constexpr int PW = 80;                                  // row pitch (bytes) of a [rows][32 features] bf16 slice
  static constexpr int KS = 0, VS = 32 * PW, Q2S = 2 * 32 * PW, DO2S = Q2S + 16 * PW, IMG = DO2S + 16 * PW, TAB = IMG + 1024, WAVE = TAB + 192;
  static constexpr int LDS = 8 * WAVE;
  static_assert(LDS <= 81920, "two blocks per CU");
};

template <bool DROP>
__global__ __launch_bounds__(512, 4) void bwd2w_kernel(const Bwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PW = CfgC::PW;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, Nk = a.Nk;
  const int head = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ws = smem + head * CfgC::WAVE;
  char* Ks = ws + CfgC::KS; char* Vs = ws + CfgC::VS; char* Q2s = ws + CfgC::Q2S; char* dO2s = ws + CfgC::DO2S; char* im = ws + CfgC::IMG;
  float* tab = reinterpret_cast<float*>(ws + CfgC::TAB);          // max [16] | 1 / sum [16] | row-dot [16] of this head's queries
  const int g0 = (int)blockIdx.x * RT;
  Sub sub[RT];
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    int4 td = make_int4(-1, 0, 0, 0);
    if (g0 + s < a.rg_tiles_max) td = a.tile_desc[g0 + s];
    const int tb = __builtin_amdgcn_readfirstlane(td.x);
    sub[s].b = tb < 0 ? 0 : tb; sub[s].row0 = tb < 0 ? 0 : __builtin_amdgcn_readfirstlane(td.y); sub[s].nr = tb < 0 ? 0 : __builtin_amdgcn_readfirstlane(td.z);
    sub[s].inv_n = 0.f;
  }
  if (sub[0].nr == 0) return;
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  u32x4 kreg[2], vreg[2], q2reg, o2reg; float t0 = 0.f, t1 = 0.f, t2 = 0.f;
  auto load_kv = [&](int row0, int nr) {                          // this head's key | value slices of 32 rows: 4 lanes per row
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = lane + 64 * i;
      const us16* p = a.KV2_16 + ((size_t)row0 + max(0, min(c >> 2, nr - 1))) * 512 + 32 * head + 8 * (c & 3);
      kreg[i] = *reinterpret_cast<const u32x4*>(p); vreg[i] = *reinterpret_cast<const u32x4*>(p + 256);
    }
  };
  auto load_sample = [&](int b) {                                 // the sample's Nk queries (pre-scaled) and d(attention output) rows, this head; its softmax tables
    const int j = lane >> 2;
    const size_t r = (size_t)b * Nk + min(j, Nk - 1);
    q2reg = *reinterpret_cast<const u32x4*>(a.Q2_16 + r * 256 + 32 * head + 8 * (lane & 3));
    o2reg = *reinterpret_cast<const u32x4*>(a.dO2_16 + r * 256 + 32 * head + 8 * (lane & 3));
    if (j >= Nk) { q2reg = u32x4{0u, 0u, 0u, 0u}; o2reg = q2reg; }
    if (lane < 16) {
      const size_t o = (size_t)b * 128 + head * 16 + lane;        // [b][head][16]
      const bool ok = lane < Nk;
      t0 = ok ? a.lse2[2 * o] : 0.f; t1 = ok ? 1.0f / a.lse2[2 * o + 1] : 0.f; t2 = ok ? a.delta2[o] : 0.f;
    }
  };
  load_kv(sub[0].row0, sub[0].nr);
  load_sample(sub[0].b);
  f32x16 dq2 = splat16(0.f);
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    if (sub[s].nr == 0) break;
    const size_t rowg0 = (size_t)sub[s].row0;
    const bool rok = l31 < sub[s].nr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = lane + 64 * i;
      *reinterpret_cast<u32x4*>(Ks + (c >> 2) * PW + 16 * (c & 3)) = kreg[i];
      *reinterpret_cast<u32x4*>(Vs + (c >> 2) * PW + 16 * (c & 3)) = vreg[i];
    }
    if (s == 0 || sub[s].b != sub[0].b) {                         // (wave-uniform) a new sample
      *reinterpret_cast<u32x4*>(Q2s + (lane >> 2) * PW + 16 * (lane & 3)) = q2reg;
      *reinterpret_cast<u32x4*>(dO2s + (lane >> 2) * PW + 16 * (lane & 3)) = o2reg;
      if (lane < 16) { tab[lane] = t0; tab[16 + lane] = t1; tab[32 + lane] = t2; }
    }
    const bool more = s + 1 < RT && sub[RT - 1].nr > 0;
    if (more) {                                                   // the next sub-tile's slices, under this one's products
      load_kv(sub[RT - 1].row0, sub[RT - 1].nr);
      if (sub[RT - 1].b != sub[0].b) load_sample(sub[RT - 1].b);
    }
    // scores of the Nk queries against this lane's row, and dP = dO2 . V2^T  ([query j][row]: lane = row, registers = queries)
    f32x16 S2 = splat16(0.f), dP = splat16(0.f);
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const int qo = (l31 & 15) * PW + 2 * (16 * ss + 8 * h), ro = l31 * PW + 2 * (16 * ss + 8 * h);
      S2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Q2s + qo), *reinterpret_cast<const bf16x8*>(Ks + ro), S2, 0, 0, 0);
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(dO2s + qo), *reinterpret_cast<const bf16x8*>(Vs + ro), dP, 0, 0, 0);
    }
    float ds[8], pd[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = acc_row(i, h);
      const float p = (j < Nk && rok) ? __expf(S2[i] - tab[j]) * tab[16 + j] : 0.f;
      const float mm = DROP ? drop_mult(a.drop, SITE_ATTN_KG2RG, ((uint32_t)(rowg0 + l31) * 8u + (uint32_t)head) * (uint32_t)Nk + (uint32_t)j) : 1.0f;
      ds[i] = p * (dP[i] * mm - tab[32 + j]);
      pd[i] = p * mm;
    }
    const u32x4 dsf = u32x4{pack2(ds[0], ds[1]), pack2(ds[2], ds[3]), pack2(ds[4], ds[5]), pack2(ds[6], ds[7])};
    const u32x4 pdf = u32x4{pack2(pd[0], pd[1]), pack2(pd[2], pd[3]), pack2(pd[4], pd[5]), pack2(pd[6], pd[7])};
    // dS image [row][16 queries] (bf16): queries 4 h .. + 3 at byte 8 h, queries 8 + 4 h .. at byte 16 + 8 h
    *reinterpret_cast<u32x2*>(im + l31 * 32 + 8 * h) = u32x2{dsf.x, dsf.y};
    *reinterpret_cast<u32x2*>(im + l31 * 32 + 16 + 8 * h) = u32x2{dsf.z, dsf.w};
    // dQ2_h[j][f] += scale * sum_rows dS2[j][row] K2[row][f]  (lane = f, registers = j): image and key slice read transposed
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const int r0 = 16 * ss + 8 * h + q4;
      const bf16x8 sA = join(lds_tr16(im + r0 * 32 + 8 * p4), lds_tr16(im + (r0 + 4) * 32 + 8 * p4));
      const int co = 2 * (16 * g1 + 4 * p4);
      const bf16x8 kB = join(lds_tr16(Ks + r0 * PW + co), lds_tr16(Ks + (r0 + 4) * PW + co));
      dq2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sA, kB, dq2, 0, 0, 0);
    }
    // dV2^T = dO2_h^T . P2d,  dK2^T = Q2_h^T . dS2: lane = row, registers = the head's 32 features
    const int tro = (4 * h + q4) * PW + 2 * (16 * g1 + 4 * p4);
    const f32x16 dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join(lds_tr16(dO2s + tro), lds_tr16(dO2s + tro + 8 * PW)), as_frag(pdf), splat16(0.f), 0, 0, 0);
    const f32x16 dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join(lds_tr16(Q2s + tro), lds_tr16(Q2s + tro + 8 * PW)), as_frag(dsf), splat16(0.f), 0, 0, 0);
    // the rows leave through the key / value slices (dead now: the wave's own LDS traffic, program order) as 16-byte pieces
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<u32x2*>(Ks + l31 * PW + 2 * (8 * g + 4 * h)) = u32x2{pack2(dk[4 * g], dk[4 * g + 1]), pack2(dk[4 * g + 2], dk[4 * g + 3])};
      *reinterpret_cast<u32x2*>(Vs + l31 * PW + 2 * (8 * g + 4 * h)) = u32x2{pack2(dv[4 * g], dv[4 * g + 1]), pack2(dv[4 * g + 2], dv[4 * g + 3])};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = lane + 64 * i, r = c >> 2;
      if (r < sub[s].nr) {
        us16* dst = a.dQKV16 + (rowg0 + r) * 768 + 256 + 32 * head + 8 * (c & 3);
        store16_wt(dst, *reinterpret_cast<const u32x4*>(Ks + r * PW + 16 * (c & 3)));
        store16_wt(dst + 256, *reinterpret_cast<const u32x4*>(Vs + r * PW + 16 * (c & 3)));
      }
    }
    if (!more || sub[RT - 1].b != sub[s].b) {                     // (wave-uniform) the sample's last sub-tile of this block: one set of atomics
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = acc_row(i, h);
        if (j < Nk) atomicAdd(a.dQ2acc + ((size_t)sub[s].b * Nk + j) * 256 + 32 * head + l31, dq2[i] * a.qscale);
      }
      dq2 = splat16(0.f);
    }
  }
}

// behind it, one block per sample: the KG rows' weight-gradient operand [dQ2 | dK | dV] (bf16) from the fp32 sums of both halves -- what
// bwd2p_kernel's early blocks and last-arriving tile blocks do
__global__ __launch_bounds__(256) void bwd2w_finish_kernel(const Bwd2Args a) {
  const size_t krow0 = (size_t)blockIdx.x * a.Nk;
  for (int c = threadIdx.x; c < a.Nk * 96; c += 256) {
    const int j = c / 96, ch = c % 96;                            // chunk of 8 values: 0 .. 31 dQ2, 32 .. 95 dK | dV
    const float* src = ch < 32 ? a.dQ2acc + (krow0 + j) * 256 + 8 * ch : a.dKV + (krow0 + j) * 512 + 8 * (ch - 32);
    const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
    *reinterpret_cast<u32x4*>(a.dQKVkg16 + (krow0 + j) * 768 + 8 * ch) = u32x4{pack2(x0.x, x0.y), pack2(x0.z, x0.w), pack2(x1.x, x1.y), pack2(x1.z, x1.w)};
  }
}

}  // namespace

// The RG rows of the backward's first half on 64-row half-blocks; the KG rows (and the KG stream's dH16) through bwd1_kernel.
int launch_wide2_bwd1(Bwd1Args& a, int variant, hipStream_t stream) {
  const int rc = launch_fused_bwd1(a, variant, stream, /*kg_only=*/1);      // validates the whole argument block
  if (rc != 0) return rc;
  const double rows = (double)a.rows_rg;
  const int prof = gemm_prof_open(stream, 2.0 * rows * (512.0 * 256.0 + 256.0 * 256.0) + 10.0 * rows * a.Nk * 256.0, PROF_BWD1);
  const Bwd1Args& k = a;
  // (k.exp == 4: developer A/B, the 4-wave shape)
  const int rc2 = k.exp == 4 ? (a.drop.p > 0.f ? bwd1w_launch<8, true, 2>(k, stream) : bwd1w_launch<8, false, 2>(k, stream))
                             : (a.drop.p > 0.f ? bwd1w_launch<6, true, 1>(k, stream) : bwd1w_launch<6, false, 1>(k, stream));
  gemm_prof_close(prof, stream);
  return rc2;
}

// The second half (parameter-space form: no dR / dG products) with the RG rows on 64-row blocks, wave = head, no barriers.
int launch_wide2_bwd2(Bwd2Args& a, hipStream_t stream) {
  if (!a.param_space) return (int)hipErrorInvalidValue;
  if (a.B < 1 || a.Nk < 1 || a.Nk > 16 || a.rg_tiles_max < 1 || !a.Q2_16 || !a.dO2_16 || !a.lse2 || !a.delta2 || !a.KV2_16 || !a.dQKV16 || !a.dQ2acc || !a.dKV ||
      !a.dQKVkg16 || !a.tile_desc)
    return (int)hipErrorInvalidValue;
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd2w_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, CfgC::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd2w_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, CfgC::LDS);
    return true;
  }();
  (void)attr;
  const int prof = gemm_prof_open(stream, 10.0 * (double)a.rows_rg * a.Nk * 256.0, PROF_BWD2);
  const dim3 grid((a.rg_tiles_max + RT - 1) / RT);
  if (a.drop.p > 0.f) hipLaunchKernelGGL(bwd2w_kernel<true>, grid, dim3(512), CfgC::LDS, stream, a);
  else                hipLaunchKernelGGL(bwd2w_kernel<false>, grid, dim3(512), CfgC::LDS, stream, a);
  hipLaunchKernelGGL(bwd2w_finish_kernel, dim3(a.B), dim3(256), 0, stream, a);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}
