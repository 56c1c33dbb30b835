// One 16-byte chunk of a weight shadow (fused_rows.h, ShadowJob): shared by shadow_kernel (fused_rows.hip) and by the extra
// blocks of the wide front kernel that build the per-sample tail's hi / lo planes (fused_wide.hip).
#pragma once
#include "fused_rows.h"

// chunk c = ((w * KS + ks) * NTw + t) * 64 + lane of job J: row n = 32 (w NTw + t) + (lane & 31), k = 16 ks + 8 (lane >> 5) + j.
// J.lo: the chunk holds bf16(W - float(bf16(W))), the second term of the two-plane split W ~ hi + lo (relative error 2^-17).
__device__ __forceinline__ void shadow_chunk(const ShadowJob& J, int c) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const int KS = J.K >> 4, NTw = J.N >> 7;
  const int lane = c & 63;
  int r = c >> 6;
  const int t = r % NTw; r /= NTw;
  const int ks = r % KS; const int w = r / KS;
  const int n = 32 * (w * NTw + t) + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
  float v[8];
  if (!J.transposed) {
    int i = 0, nn = n;
    while (i + 1 < J.nsrc && nn >= J.rows[i]) { nn -= J.rows[i]; ++i; }
    // (two 16-byte loads: eight dword loads of 64 different cache lines each kept the address unit busy 4x longer)
    const float4* p = reinterpret_cast<const float4*>(J.src[i] + (size_t)nn * J.ld[i] + k0);
    const float4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    int i = 0, kk = k0;
    while (i + 1 < J.nsrc && kk >= J.rows[i]) { kk -= J.rows[i]; ++i; }      // (source blocks are multiples of 8 rows)
    const float* p = J.src[i] + (size_t)kk * J.ld[i] + n;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = p[(size_t)e * J.ld[i]];
  }
  if (J.lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] -= __uint_as_float((uint32_t)f2bf(v[e]) << 16);
  }
  reinterpret_cast<u32x4_t*>(J.dst)[c] = u32x4_t{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
}
