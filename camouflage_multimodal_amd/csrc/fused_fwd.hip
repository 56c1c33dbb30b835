// Fused node-level forward (hidden 256, 8 heads, bf16 MFMA): one workgroup carries a 32-node
// slice of one sample through
//   rg_proj -> {Q | K2 | V2} in-projections -> RG->KG attention -> out-proj + residual -> LayerNorm
//   -> FFN layer 0 (+ReLU, dropout) -> partial column sums for the mean-pool
// in ONE launch.  The unfused schedule ran this as 4 GEMM launches + attention + LayerNorm + pooling,
// each paying a launch floor, prologue/epilogue round trips and a re-read of its input from the
// Infinity Cache (the producer's XCD L2 is not the consumer's).  Here a stage's output tile stays in
// LDS as the bf16 A operand of the next stage; only what backward needs is written to HBM (same
// workspace buffers as the unfused path, so backward and the parity tests are unchanged).
//
// GEMM stages: C[64 x 256] chunks, 4 waves as 2(M) x 2(N), each 32 x 128 = four 32x32x16 bf16 MFMA
// accumulators sharing the A fragment.  A = activation tile in LDS ([64][K+8] bf16: 16-lane b128 reads
// hit 16 distinct slots).  B = weight K-tiles [256][32] streamed from the bf16 shadow copy (0.85 MB,
// L2-resident) through registers with FOUR tiles in flight into a double-buffered 80-byte-pitch LDS
// image; one barrier per K-tile.  Per K-tile a CU does 8 MFMAs per wave (256 cycles) against a 16 KB
// weight fetch (256 cycles at 64 B/clk): balanced by construction.
#include "fused.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned short us;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // first-class 16-B value (HIP's uint4 is a struct)

constexpr int RT = 32, H = 256, WPITCH = 40, PD = 4;   // 32 nodes per workgroup: ~240 workgroups at B = 16, one per CU

__device__ __forceinline__ us f2bf(float f) { __bf16 b = (__bf16)f; return __builtin_bit_cast(us, b); }

struct Lane {
  int tid, lane, wave, l31, h;
  __device__ Lane() { tid = threadIdx.x; lane = tid & 63; wave = tid >> 6; l31 = lane & 31; h = lane >> 5; }
  // The 4 waves split the 256 output features of a chunk (64 each = two 32x32 tiles); all cover the same 32
  // nodes.  Transposed accumulator tile j (32 features x 32 nodes): this lane's node is l31, registers
  // 4g..4g+3 hold the 4 consecutive features starting at feat(j, g).
  __device__ __forceinline__ int node() const { return l31; }
  __device__ __forceinline__ int feat(int j, int g) const { return 64 * wave + 32 * j + 8 * g + 4 * h; }
};
constexpr int NT = 2;   // accumulator tiles per wave

// acc[4] (+)= As[64][KDIM] . W[256][KDIM]^T   (W: one 256-row block of the pre-tiled bf16 shadow weights)
template <int KDIM>
__device__ __forceinline__ void tile_gemm(const Lane& L, const us* As, const us* __restrict__ W, us* Wt, f32x16 (&acc)[NT]) {
  constexpr int NKT = KDIM / 32, AP = KDIM + 8;
  static_assert(NKT % PD == 0, "K tiles must be a multiple of the prefetch depth");
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // four register stages as NAMED objects: an array of stages indexed through a lambda parameter was
  // not promoted to registers by hipcc (the weight tiles went global -> scratch -> LDS).
  struct WStage { u32x4 v0, v1, v2, v3; };
  WStage s0, s1, s2, s3;
  // The shadow weights are stored PRE-TILED (cast_tiled_bf16_kernel): K-tile kt of this 256-row block is
  // one contiguous 16 KB run [256 rows][32 k], so a wave-instruction reads 1 KB of consecutive bytes
  // (a row-major source gave each lane its own 128-B line: 8x the L2->L1 traffic, ~2 us per K-tile).
  // 16-B chunk c = tid + 256 i  ->  row c >> 2, k-chunk c & 3.
  auto gl = [&](WStage& r, int kt) {
    const u32x4* p = reinterpret_cast<const u32x4*>(W + (size_t)kt * (256 * 32)) + L.tid;
    r.v0 = p[0]; r.v1 = p[256]; r.v2 = p[512]; r.v3 = p[768];
  };
  auto st = [&](const WStage& r, int buf) {
    us* base = Wt + buf * (256 * WPITCH) + (L.tid >> 2) * WPITCH + (L.tid & 3) * 8;
    *reinterpret_cast<u32x4*>(base) = r.v0;
    *reinterpret_cast<u32x4*>(base + 64 * WPITCH) = r.v1;
    *reinterpret_cast<u32x4*>(base + 128 * WPITCH) = r.v2;
    *reinterpret_cast<u32x4*>(base + 192 * WPITCH) = r.v3;
  };
  auto compute = [&](int buf, int kt) {
    const us* ap = As + L.l31 * AP + 32 * kt + 8 * L.h;
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ap), a1 = *reinterpret_cast<const bf16x8*>(ap + 16);
    const us* bp = Wt + buf * (256 * WPITCH) + (64 * L.wave + L.l31) * WPITCH + 8 * L.h;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bp + 32 * j * WPITCH);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bp + 32 * j * WPITCH + 16);
      // weights as the A operand: the accumulator tile is C^T -- rows (registers) = output features,
      // col (lane) = node -- so a lane owns 4 CONSECUTIVE features of one node per register quad and the
      // epilogues store 16 B per instruction (dword stores made every epilogue ~9 us: store-issue bound)
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[j], 0, 0, 0);
    }
  };
  gl(s0, 0); gl(s1, 1); gl(s2, 2); gl(s3, 3);
  st(s0, 0);
  if (PD < NKT) gl(s0, PD);
  __syncthreads();
  // tile kt sits in LDS buffer kt & 1; while it is multiplied, tile kt+1 (loaded 3 iterations ago) is
  // written to the other buffer and its stage refilled with tile kt+5.  One barrier per tile.
#pragma unroll 1
  for (int t = 0; t < NKT; t += PD) {
    compute(0, t);
    st(s1, 1); if (t + 5 < NKT) gl(s1, t + 5);
    __syncthreads();
    compute(1, t + 1);
    st(s2, 0); if (t + 6 < NKT) gl(s2, t + 6);
    __syncthreads();
    compute(0, t + 2);
    st(s3, 1); if (t + 7 < NKT) gl(s3, t + 7);
    __syncthreads();
    compute(1, t + 3);
    if (t + 4 < NKT) { st(s0, 0); if (t + 8 < NKT) gl(s0, t + 8); }
    __syncthreads();
  }
}

// ---- 16x16x4 f32 MFMA helpers of the attention stage (see attn_mfma.hip for the operand algebra)
struct Frag8 { float4 lo, hi; };
__device__ __forceinline__ Frag8 load_row8(const float* __restrict__ row, int q, bool valid) {
  Frag8 f;
  f.lo = *reinterpret_cast<const float4*>(row + 8 * q);
  f.hi = *reinterpret_cast<const float4*>(row + 8 * q + 4);
  if (!valid) { f.lo = make_float4(0.f, 0.f, 0.f, 0.f); f.hi = f.lo; }
  return f;
}
__device__ __forceinline__ f4 mma_nt32(const Frag8& a, const Frag8& b, f4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.x, b.lo.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.y, b.lo.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.z, b.lo.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.w, b.lo.w, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.x, b.hi.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.y, b.hi.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.z, b.hi.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.w, b.hi.w, c, 0, 0, 0);
  return c;
}
__device__ __forceinline__ f4 mma_acc16(const f4& a, const f4& b, f4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
  return c;
}
__device__ __forceinline__ float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
// sum over the 32 lanes that share lane>>5 (the 32 columns of one MFMA tile row)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256, 1) void rg_forward_fused_kernel(const RgFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Lane L;
  const int b = blockIdx.y;
  const int r0 = a.offs[b], nr = a.offs[b + 1] - r0;
  const int t0 = blockIdx.x * RT;
  if (t0 >= nr) return;
  const int rows = min(RT, nr - t0);
  const int D = a.D, XP = D + 8;
  us* Xs = reinterpret_cast<us*>(smem);              // [64][D+8]
  us* Rs = Xs + RT * (256 + 8);                      // [64][264]   (Xs region sized for D <= 256)
  us* Os = Rs + RT * 264;                            // [64][264]   O, later Y
  us* Wt = Os + RT * 264;                            // [2][256][40]
  float* red = reinterpret_cast<float*>(Wt + 2 * 256 * WPITCH);   // [32][4]
  f32x16 acc[NT];
  const float inv_nr = 1.0f / (float)nr;

  // ---- S0: X tile -> bf16
  for (int i = L.tid; i < RT * (D / 4); i += 256) {
    const int r = i / (D / 4), c = (i - r * (D / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) v = *reinterpret_cast<const float4*>(a.X + (size_t)(r0 + t0 + r) * D + c);
    *reinterpret_cast<uint2*>(Xs + r * XP + c) = make_uint2((uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16),
                                                           (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16));
  }
  // (tile_gemm's first barrier orders these stores before the first fragment read)

  // ---- S1: R = X.Wrg^T + b
  if (D == 128) tile_gemm<128>(L, Xs, a.Wrg, Wt, acc);
  else if (D == 256) tile_gemm<256>(L, Xs, a.Wrg, Wt, acc);
  else tile_gemm<512>(L, Xs, a.Wrg, Wt, acc);      // unreachable (launcher admits D in {128, 256})
  const int nd = L.node();
  const bool nd_ok = nd < rows;
  const size_t trow = (size_t)(r0 + t0 + min(nd, rows - 1));
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = L.feat(j, g);
      const float4 bv = *reinterpret_cast<const float4*>(a.brg + f);
      const float4 v = make_float4(acc[j][4 * g] + bv.x, acc[j][4 * g + 1] + bv.y, acc[j][4 * g + 2] + bv.z, acc[j][4 * g + 3] + bv.w);
      if (nd_ok) *reinterpret_cast<float4*>(a.R + trow * H + f) = v;
      *reinterpret_cast<uint2*>(Rs + nd * 264 + f) = make_uint2((uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16),
                                                               (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16));
    }
  if (a.debug_stop == 1) return;
  // ---- S2: Q | K2 | V2 = R.W^T + b
#pragma unroll 1
  for (int n = 0; n < 3; ++n) {
    const us* W = n == 0 ? a.Wq : a.Wkv2 + (size_t)(n - 1) * 256 * 256;
    const float* bias = n == 0 ? a.bq : a.bkv2 + (n - 1) * 256;
    float* dst = n == 0 ? a.Q + trow * H : a.KV2 + trow * 2 * H + (n - 1) * H;
    tile_gemm<256>(L, Rs, W, Wt, acc);
    if (a.debug_stop == 5) { if (acc[0][0] == 123.456f) a.Q[0] = acc[1][1]; return; }
    if (nd_ok) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f = L.feat(j, g);
          const float4 bv = *reinterpret_cast<const float4*>(bias + f);
          *reinterpret_cast<float4*>(dst + f) =
              make_float4(acc[j][4 * g] + bv.x, acc[j][4 * g + 1] + bv.y, acc[j][4 * g + 2] + bv.z, acc[j][4 * g + 3] + bv.w);
        }
    }
    if (a.debug_stop == 6) return;
  }
  if (a.debug_stop == 2) return;
  __syncthreads();   // Q rows of this tile are complete (same-CU global writes, read back below)

  // ---- S3: RG->KG attention; wave w owns nodes 16(w&1)..+15 of the tile and half of the heads
  {
    const int x = L.lane & 15, q = L.lane >> 4;
    const int s0 = 16 * (L.wave & 1);
    const int nh = a.nh, Nk = a.Nk;
    const int h_beg = (L.wave >> 1) * (nh >> 1), h_end = h_beg + (nh >> 1);
    const int node = r0 + min(t0 + s0 + x, nr - 1);
    const bool node_ok = t0 + s0 + x < nr;
    const float* kvb = a.KV + (size_t)b * Nk * 2 * H;
    for (int hh = h_beg; hh < h_end; ++hh) {
      const Frag8 qf = load_row8(a.Q + (size_t)node * H + hh * 32, q, true);
      const Frag8 kf = load_row8(kvb + (size_t)min(x, Nk - 1) * 2 * H + hh * 32, q, x < Nk);
      f4 vb[2];
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
        for (int e = 0; e < 4; ++e) vb[n2][e] = kvb[(size_t)min(4 * q + e, Nk - 1) * 2 * H + H + hh * 32 + 16 * n2 + x];
      f4 s = mma_nt32(kf, qf, f4{0.f, 0.f, 0.f, 0.f});     // S^T: rows = keys 4q+r, col = node x
      float m = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[r] = (4 * q + r < Nk) ? s[r] * a.scale : -INFINITY; m = fmaxf(m, s[r]); }
      m = group_max(m);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[r] = __expf(s[r] - m); sum += s[r]; }
      sum = group_sum(sum);
      const float inv = 1.0f / sum;
      const size_t pbase = ((size_t)node * nh + hh) * Nk;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 4 * q + r;
        float p = s[r] * inv;
        if (key < Nk) {
          if (node_ok) a.P[pbase + key] = p;
          if (a.drop.p > 0.f) p *= drop_mult(a.drop, SITE_ATTN_RG2KG, (uint32_t)(pbase + key));
        }
        s[r] = p;
      }
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) {
        const f4 o = mma_acc16(s, vb[n2], f4{0.f, 0.f, 0.f, 0.f});   // rows = nodes 4q+r, col = 16 n2 + x
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rw = s0 + 4 * q + r, c = hh * 32 + 16 * n2 + x;
          if (rw < rows) a.O[(size_t)(r0 + t0 + rw) * H + c] = o[r];
          Os[rw * 264 + c] = f2bf(o[r]);
        }
      }
    }
  }
  if (a.debug_stop == 3) return;
  // ---- S4: U = R + O.Wo^T + b ; LayerNorm -> Y  (tile_gemm's first barrier publishes Os)
  tile_gemm<256>(L, Os, a.Wo, Wt, acc);
  {
    // this lane holds 32 of its node's 256 features (wave, half h): row statistics are an in-lane sum, one
    // xor-32 shuffle and an exchange between the four waves through LDS
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f = L.feat(j, g);
        const float4 bv = *reinterpret_cast<const float4*>(a.bo + f);
        const float4 rv = *reinterpret_cast<const float4*>(a.R + trow * H + f);   // this lane's own S1 stores
        acc[j][4 * g] += bv.x + rv.x; acc[j][4 * g + 1] += bv.y + rv.y; acc[j][4 * g + 2] += bv.z + rv.z; acc[j][4 * g + 3] += bv.w + rv.w;
        if (nd_ok) *reinterpret_cast<float4*>(a.U + trow * H + f) = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
        s1 += (acc[j][4 * g] + acc[j][4 * g + 1]) + (acc[j][4 * g + 2] + acc[j][4 * g + 3]);
      }
    s1 += __shfl_xor(s1, 32, 64);
    if (L.h == 0) red[nd * 4 + L.wave] = s1;
    __syncthreads();
    const float mean = ((red[nd * 4] + red[nd * 4 + 1]) + (red[nd * 4 + 2] + red[nd * 4 + 3])) * (1.0f / H);
    __syncthreads();
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = acc[j][r] - mean; s2 = fmaf(d, d, s2); }
    s2 += __shfl_xor(s2, 32, 64);
    if (L.h == 0) red[nd * 4 + L.wave] = s2;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[nd * 4] + red[nd * 4 + 1]) + (red[nd * 4 + 2] + red[nd * 4 + 3])) * (1.0f / H) + 1e-5f);
    if (L.wave == 0 && L.h == 0 && nd_ok) { a.stats[trow * 2] = mean; a.stats[trow * 2 + 1] = rstd; }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f = L.feat(j, g);
        const float4 gm = *reinterpret_cast<const float4*>(a.ln_g + f), bt = *reinterpret_cast<const float4*>(a.ln_b + f);
        float4 y = make_float4((acc[j][4 * g] - mean) * rstd * gm.x + bt.x, (acc[j][4 * g + 1] - mean) * rstd * gm.y + bt.y,
                               (acc[j][4 * g + 2] - mean) * rstd * gm.z + bt.z, (acc[j][4 * g + 3] - mean) * rstd * gm.w + bt.w);
        if (nd_ok) *reinterpret_cast<float4*>(a.Y + trow * H + f) = y;
        *reinterpret_cast<uint2*>(Os + nd * 264 + f) = make_uint2((uint32_t)f2bf(y.x) | ((uint32_t)f2bf(y.y) << 16),
                                                                 (uint32_t)f2bf(y.z) | ((uint32_t)f2bf(y.w) << 16));   // Y tile = A of the FFN stage
        // pooling partial: column sums over the wave's 32 nodes
        if (!nd_ok) y = make_float4(0.f, 0.f, 0.f, 0.f);
        const float c0 = half_sum(y.x), c1 = half_sum(y.y), c2 = half_sum(y.z), c3 = half_sum(y.w);
        if (L.l31 == 0) {
          float* dm = a.Ymean + (size_t)b * H + f;
          atomicAdd(dm, c0 * inv_nr); atomicAdd(dm + 1, c1 * inv_nr); atomicAdd(dm + 2, c2 * inv_nr); atomicAdd(dm + 3, c3 * inv_nr);
        }
      }
  }
  if (a.debug_stop == 4) return;
  // ---- S5: H1 = dropout(relu(Y.W1^T + b)), 2 x 256 columns
#pragma unroll 1
  for (int n = 0; n < 2; ++n) {
    tile_gemm<256>(L, Os, a.W1 + (size_t)n * 256 * 256, Wt, acc);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f = n * 256 + L.feat(j, g);
        const float4 bv = *reinterpret_cast<const float4*>(a.b1 + f);
        float4 v = make_float4(fmaxf(acc[j][4 * g] + bv.x, 0.f), fmaxf(acc[j][4 * g + 1] + bv.y, 0.f),
                               fmaxf(acc[j][4 * g + 2] + bv.z, 0.f), fmaxf(acc[j][4 * g + 3] + bv.w, 0.f));
        if (a.drop.p > 0.f) {
          const uint32_t idx = (uint32_t)trow * (uint32_t)(2 * H) + (uint32_t)f;
          v.x *= drop_mult(a.drop, SITE_FFN_RG, idx); v.y *= drop_mult(a.drop, SITE_FFN_RG, idx + 1);
          v.z *= drop_mult(a.drop, SITE_FFN_RG, idx + 2); v.w *= drop_mult(a.drop, SITE_FFN_RG, idx + 3);
        }
        if (nd_ok) *reinterpret_cast<float4*>(a.H1 + trow * 2 * H + f) = v;
        else v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float c0 = half_sum(v.x), c1 = half_sum(v.y), c2 = half_sum(v.z), c3 = half_sum(v.w);
        if (L.l31 == 0) {
          float* dm = a.H1mean + (size_t)b * 2 * H + f;
          atomicAdd(dm, c0 * inv_nr); atomicAdd(dm + 1, c1 * inv_nr); atomicAdd(dm + 2, c2 * inv_nr); atomicAdd(dm + 3, c3 * inv_nr);
        }
      }
  }
}

// fp32 [N][K] row-major -> bf16 tiled [N/256][K/32][256][32]  (N % 256 == 0, K % 32 == 0)
struct CastJobs { const float* src[8]; us* dst[8]; int N[8]; int K[8]; int count; };
__global__ __launch_bounds__(256) void cast_tiled_bf16_kernel(const CastJobs jobs) {
  const int jb = blockIdx.y;
  if (jb >= jobs.count) return;
  const int N = jobs.N[jb], K = jobs.K[jb];
  const float* src = jobs.src[jb];
  us* dst = jobs.dst[jb];
  const int n4 = (N * K) >> 2;                       // one thread-iteration = 4 consecutive k of one row
  const int k4 = K >> 2, kt_n = K >> 5;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
    const int row = i / k4, k = (i - row * k4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)row * K + k);
    const size_t o = (((size_t)(row >> 8) * kt_n + (k >> 5)) * 256 + (row & 255)) * 32 + (k & 31);
    *reinterpret_cast<uint2*>(dst + o) = make_uint2((uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16),
                                                    (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16));
  }
}

}  // namespace

size_t rg_fused_lds_bytes() {
  return sizeof(us) * ((size_t)RT * 264 * 3 + 2 * 256 * WPITCH) + sizeof(float) * RT * 4;
}

int rg_fused_supported(int D, int Hd, int nh, int Nk) {
  return Hd == 256 && nh == 8 && (D == 128 || D == 256) && Nk >= 1 && Nk <= 16;
}

int launch_cast_tiled_bf16(const float* const* src, unsigned short* const* dst, const int* N, const int* K, int count,
                           hipStream_t stream) {
  if (count < 1 || count > 8) return (int)hipErrorInvalidValue;
  CastJobs j{};
  j.count = count;
  for (int i = 0; i < count; ++i) {
    if ((N[i] & 255) || (K[i] & 31)) return (int)hipErrorInvalidValue;
    j.src[i] = src[i]; j.dst[i] = dst[i]; j.N[i] = N[i]; j.K[i] = K[i];
  }
  hipLaunchKernelGGL(cast_tiled_bf16_kernel, dim3(32, count), dim3(256), 0, stream, j);
  return (int)hipGetLastError();
}

int launch_rg_forward_fused(const RgFwdArgs& a, int B, int max_nr, hipStream_t stream) {
  const size_t lds = rg_fused_lds_bytes();
  (void)hipFuncSetAttribute((const void*)rg_forward_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(rg_forward_fused_kernel, dim3((max_nr + RT - 1) / RT, B), dim3(256), lds, stream, a);
  return (int)hipGetLastError();
}
