// mfma_power -- what the matrix pipe itself sustains under this part's power limit, per MFMA shape, on realistic operand values.
//
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/mfma_power.cpp -o tools/bin/mfma_power && tools/bin/mfma_power [seconds = 1.0]
//
// One wave per SIMD (256 threads per CU, one workgroup per CU, 512 registers each like the four-wave GEMM), operands resident in
// registers, nothing but MFMAs in the loop: v_mfma_f32_16x16x32_bf16 on 64 accumulator quads against v_mfma_f32_32x32x16_bf16 on 16
// accumulator blocks -- the same 128 x 128 wave tile, the same FLOPs per K step. Operands ~ N(0,1) / N(0,0.02^2) (zeros run at a higher
// clock: DESIGN.md 3.1), windows of the two shapes alternate. If the two shapes sustain the same TFLOP/s, the shape is not a power lever
// for the GEMM; if 32x32x16 (half the operand register reads per FLOP) sustains more, it is. Measurement helper, not product code.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                        \
      exit(2);                                                                       \
    }                                                                                \
  } while (0)

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 in 4 VGPRs
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// 128 x 128 x 64 per iteration: 8 A fragments x 8 B fragments x 2 k-halves of 16x16x32 = 128 MFMAs
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k16(const bf16x8* __restrict__ a, const bf16x8* __restrict__ b, float* out, int iters) {
  const int t = threadIdx.x + blockIdx.x * 256;
  bf16x8 fa[2][8], fb[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      fa[h][i] = a[(t * 16 + h * 8 + i) & 0xffff];
      fb[h][i] = b[(t * 16 + h * 8 + i) & 0xffff];
    }
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[h][i], fb[h][j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[t] = s;
}

// the same tile and FLOPs: 4 A fragments (32 rows x 16 k) x 4 B fragments x 4 k-quarters of 32x32x16 = 64 MFMAs
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k32(const bf16x8* __restrict__ a, const bf16x8* __restrict__ b, float* out, int iters) {
  const int t = threadIdx.x + blockIdx.x * 256;
  bf16x8 fa[4][4], fb[4][4];
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[h][i] = a[(t * 16 + h * 4 + i) & 0xffff];
      fb[h][i] = b[(t * 16 + h * 4 + i) & 0xffff];
    }
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][i], fb[h][j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
  out[t] = s;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 1.0;
  int ncu = 0;
  CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  std::mt19937 rng(1);
  std::normal_distribution<float> na(0.f, 1.f), nb(0.f, 0.02f);
  std::vector<uint16_t> ha(65536 * 8), hb(65536 * 8);
  for (auto& v : ha) v = f2bf(na(rng));
  for (auto& v : hb) v = f2bf(nb(rng));
  bf16x8 *da, *db;
  float* dout;
  CK(hipMalloc(&da, ha.size() * 2));
  CK(hipMalloc(&db, hb.size() * 2));
  CK(hipMalloc(&dout, (size_t)ncu * 256 * 4));
  CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  const int iters = 20000;                                        // 128 x 128 x 64 x 2 FLOP per wave and iteration
  const double flop = 2.0 * 128 * 128 * 64 * iters * 4.0 * ncu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int round = 0; round < 3; ++round)
    for (int which = 0; which < 2; ++which) {
      double best = 1e30, sum = 0;
      int n = 0;
      const auto t0 = std::chrono::steady_clock::now();
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0));
        if (which == 0) hipLaunchKernelGGL(k16, dim3(ncu), dim3(256), 0, 0, da, db, dout, iters);
        else hipLaunchKernelGGL(k32, dim3(ncu), dim3(256), 0, 0, da, db, dout, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min<double>(best, ms);
        sum += ms;
        ++n;
      }
      printf("{\"round\": %d, \"mfma\": \"%s\", \"launches\": %d, \"mean_ms\": %.3f, \"min_ms\": %.3f, \"sustained_tflops\": %.1f, \"cus\": %d}\n", round,
             which == 0 ? "16x16x32_bf16" : "32x32x16_bf16", n, sum / n, best, flop / (sum / n * 1e-3) / 1e12, ncu);
      fflush(stdout);
    }
  return 0;
}
