// Developer microbenchmark (not part of the product): how fast can ONE CU stream bf16 weight fragments out of L2 in the
// access shape of csrc/fused_rows.hip (one 16-byte load per lane = 1 KB per wave-instruction, DEPTH of them in flight per
// wave, every block streaming the SAME buffer so that it is L2-resident), as a function of waves per CU and DEPTH?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_bench tools/dev/microbench/stream_bench.hip && /tmp/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ void stream_kernel(const u32x4* __restrict__ w, int frags_per_wave, int nwaves_total_stride, unsigned int* sink) {
  extern __shared__ char lds[];                            // (only to pin the number of blocks per CU)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4* p = w + (size_t)wave * frags_per_wave * 64 + lane;
  u32x4 f[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) f[i] = p[(size_t)i * 64];
  unsigned int acc = 0;
  int n = DEPTH;
  for (; n + DEPTH <= frags_per_wave; n += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      acc ^= f[i].x ^ f[i].y ^ f[i].z ^ f[i].w;
      f[i] = p[(size_t)(n + i) * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) acc ^= f[i].x ^ f[i].y ^ f[i].z ^ f[i].w;
  if (acc == 0x12345678u) sink[0] = acc + (unsigned)lds[threadIdx.x];
}

// the same stream feeding MFMAs the way Stage<NT, KS>::run does: per k step one activation fragment from LDS, NT weight
// fragments -> NT MFMAs (v_mfma_f32_32x32x16_bf16), prefetch ring DEPTH deep, sched_barrier per step
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NT, int DEPTH>
__global__ __launch_bounds__(256, 2) void stage_kernel(const u32x4* __restrict__ w, int ksteps, float* sink) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32 * 528 / 4; i += blockDim.x) reinterpret_cast<unsigned int*>(lds)[i] = 0x3F803F80u;
  __syncthreads();
  const u32x4* p = w + (size_t)wave * ksteps * NT * 64 + lane;
  constexpr int AHEAD = DEPTH / NT;                        // k steps in flight
  u32x4 f[AHEAD][NT];
#pragma unroll
  for (int a = 0; a < AHEAD; ++a)
#pragma unroll
    for (int t = 0; t < NT; ++t) f[a][t] = p[(size_t)(a * NT + t) * 64];
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const char* act = lds + (lane & 31) * 528 + 16 * (lane >> 5);
  for (int ks = 0; ks < ksteps; ks += AHEAD) {
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) {
      const u32x4 av = *reinterpret_cast<const u32x4*>(act + 32 * ((ks + a) & 15));
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[a][t]), __builtin_bit_cast(bf16x8, av), acc[t], 0, 0, 0);
        const int nk = ks + a + AHEAD;
        f[a][t] = p[(size_t)((nk < ksteps ? nk : ksteps - 1) * NT + t) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][7];
  if (s == 123.456f) sink[0] = s;
}

template <int NT, int DEPTH>
static void run_stage(int ksteps, const u32x4* buf, float* sink, int cus, int blocks_per_cu) {
  const size_t lds = blocks_per_cu == 1 ? 100 * 1024 : 70 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<NT, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = cus * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stage_kernel<NT, DEPTH>), dim3(grid), dim3(256), lds, 0, buf, ksteps, sink);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stage_kernel<NT, DEPTH>), dim3(grid), dim3(256), lds, 0, buf, ksteps, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, bytes = 4.0 * ksteps * NT * 1024;
  printf("stage NT %d  k steps %3d  depth %2d  blocks/CU %d  %4.0f KB/block: %7.2f us/launch  %6.1f GB/s per CU  (MFMA time alone %.2f us)\n", NT, ksteps, DEPTH,
         blocks_per_cu, bytes / 1024, us, bytes * blocks_per_cu / (us * 1e-6) / 1e9, ksteps * NT * 32 / 2400.0);
}


// Cold vs warm: in the product every node-level kernel of a B = 16 step meets its weights for the first time since the previous
// step (307 MB of traffic ago: gone from the 4 MB L2s, still in the Infinity Cache).  thrash_kernel evicts the L2s between
// launches; prefetch_kernel is what a preceding kernel's last blocks could do: every XCD (block b runs on XCD b % 8) reads the
// whole buffer once, 1/32 per block.
__global__ void thrash_kernel(u32x4* __restrict__ t, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    u32x4 v = t[i]; v.x += 1; t[i] = v;
  }
}
__global__ void prefetch_kernel(const u32x4* __restrict__ w, size_t n16, unsigned int* sink) {
  const int part = blockIdx.x >> 3, parts = gridDim.x >> 3;
  const size_t per = (n16 + parts - 1) / parts, lo = part * per, hi = lo + per < n16 ? lo + per : n16;
  unsigned int acc = 0;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) { const u32x4 v = w[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int NT, int DEPTH>
static void run_stage_cold(int ksteps, const u32x4* buf, float* sink, int cus, u32x4* trash, size_t trash_n) {
  const size_t lds = 100 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<NT, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const size_t bytes = (size_t)4 * ksteps * NT * 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double us[4];
  for (int mode = 0; mode < 4; ++mode) {                    // 0 thrash; 1 thrash + stage; 2 thrash + prefetch; 3 thrash + prefetch + stage
    const int reps = 20;
    for (int i = -3; i < reps; ++i) {
      if (i == 0) hipEventRecord(e0);
      hipLaunchKernelGGL(thrash_kernel, dim3(1024), dim3(256), 0, 0, trash, trash_n);
      if (mode >= 2) hipLaunchKernelGGL(prefetch_kernel, dim3(256), dim3(256), 0, 0, buf, bytes / 16, reinterpret_cast<unsigned int*>(sink));
      if (mode & 1) hipLaunchKernelGGL((stage_kernel<NT, DEPTH>), dim3(cus), dim3(256), lds, 0, buf, ksteps, sink);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    us[mode] = ms * 1e3 / reps;
  }
  printf("cold stage NT %d  k steps %3d  %4zu KB/block: thrash %.2f us; stage after thrash %.2f us (%.1f GB/s per CU); prefetch %.2f us; stage after prefetch %.2f us (%.1f GB/s per CU)\n",
         NT, ksteps, bytes / 1024, us[0], us[1] - us[0], bytes / ((us[1] - us[0]) * 1e-6) / 1e9, us[2] - us[0], us[3] - us[2],
         bytes / ((us[3] - us[2]) * 1e-6) / 1e9);
}

template <int DEPTH>
static void run(int waves_per_block, int blocks_per_cu, size_t bytes_per_block, const u32x4* buf, unsigned int* sink, int cus) {
  const int threads = 64 * waves_per_block;
  const int frags_per_wave = (int)(bytes_per_block / 1024 / waves_per_block);
  const size_t lds = blocks_per_cu == 1 ? 100 * 1024 : (blocks_per_cu == 2 ? 70 * 1024 : 36 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = cus * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(grid), dim3(threads), lds, 0, buf, frags_per_wave, 0, sink);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(grid), dim3(threads), lds, 0, buf, frags_per_wave, 0, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double per_cu = (double)bytes_per_block * blocks_per_cu / (us * 1e-6) / 1e9;
  printf("waves/block %2d  blocks/CU %d  depth %2d  %4zu KB/block: %7.2f us/launch  %6.1f GB/s per CU  %6.2f TB/s chip  (%d KB in flight per CU)\n",
         waves_per_block, blocks_per_cu, DEPTH, bytes_per_block / 1024, us, per_cu, per_cu * cus / 1e3, waves_per_block * blocks_per_cu * DEPTH);
}

int main() {
  int dev = 0, cus = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const size_t bytes = 3 * 1024 * 1024;                    // the largest stream below, L2-resident on every XCD after the warm-up launches
  u32x4* buf; unsigned int* sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 4); hipMemset(buf, 1, bytes);
  printf("%d CUs; every block streams the same buffer (weights of a layer chain), fragment order, 16 B per lane\n", cus);
  for (size_t kb : {384, 1536}) {
    const size_t b = kb * 1024;
    run<12>(4, 1, b, buf, sink, cus);
    run<24>(4, 1, b, buf, sink, cus);
    run<12>(8, 1, b, buf, sink, cus);
    run<12>(16, 1, b, buf, sink, cus);
    run<6>(16, 1, b, buf, sink, cus);
    run<12>(4, 2, b, buf, sink, cus);
    run<12>(4, 4, b, buf, sink, cus);
  }
  float* fs; hipMalloc(&fs, 4);
  run_stage<6, 12>(16, buf, fs, cus, 1);    // front stage 1: 256 -> 768
  run_stage<6, 12>(64, buf, fs, cus, 1);    // (four times as long: the steady state without the launch)
  run_stage<6, 24>(64, buf, fs, cus, 1);
  run_stage<2, 12>(64, buf, fs, cus, 1);    // out-projection shape, long
  run_stage<4, 12>(64, buf, fs, cus, 1);    // FFN shape, long
  run_stage<6, 12>(64, buf, fs, cus, 2);
  u32x4* trash; const size_t trash_bytes = (size_t)96 << 20; hipMalloc(&trash, trash_bytes); hipMemset(trash, 0, trash_bytes);
  run_stage_cold<6, 12>(16, buf, fs, cus, trash, trash_bytes / 16);
  run_stage_cold<6, 12>(64, buf, fs, cus, trash, trash_bytes / 16);
  run_stage_cold<4, 12>(64, buf, fs, cus, trash, trash_bytes / 16);
  return 0;
}
