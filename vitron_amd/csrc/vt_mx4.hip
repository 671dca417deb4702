// vt_mx4.hip -- MX-FP4 images of GEMM operands for precise level 3.
//
// The reference computes every Linear in one 16-bit format (vitron/model/builder.py:47: torch_dtype = float16) and north_star asks for
// logits within 1e-3 of its fp32 CPU path; the 16-bit store of every GEMM A operand is what carries a full-depth prefill past that
// (profiles/r5_parity_round_points_fulldepth.txt). Level 3 keeps the 16-bit operand on the 16-bit MFMA and adds the rounding
// remainder lo = v - f32(op16(v)) -- about 2^-12 of the value, so 3..4 significant bits of it are enough -- on the MX pipe at four
// times the rate: v_mfma_scale_f32_16x16x128_f8f6f4 with e2m1 elements and e8m0 block scales, into the SAME fp32 accumulator
// (vt_gemm8x.inc). This file makes the operands of that instruction:
//   weights      W [N][K] 16 bit  ->  W4 [N][K/2] bytes (element k in byte k/2, even k in bits 3:0) + ONE biased exponent per row;
//   activations  lo [M][K]        ->  A4 [M][K/2] bytes + one biased exponent per (row, 32 consecutive k), stored where the GEMM's
//                                      lanes fetch them: aexp[((m / 64) * (K / 32) + kb) * 64 + (m % 16) * 4 + (m % 64) / 16]
//                                      (a lane of the MFMA holds row m % 16 of four 16-row fragments: one dword = its four scales);
// from a 16-bit lo (the precise level 2 kernels' second output), or fused into RMSNorm / LayerNorm, which have the fp32 value in hand.
// Quantisation (oracle/vitron_oracle.py mx4_exponent / mx4_round restate it, tests compare bit for bit): scale 2^e with e the
// smallest exponent for which the block's largest magnitude is <= 6 (nothing clips), elements to the nearest e2m1 value, ties to the
// even mantissa. tools/mx_probe.hip pins the instruction's operand layout and scale semantics on the hardware.
#include "vt_common.h"
#include "vt_kernels.h"
#include "vt_mx4.h"

namespace {

__device__ __forceinline__ float quad_max(float v) {   // over the 4 lanes of a quad
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  return fmaxf(v, __shfl_xor(v, 2, 64));
}
// one wave per weight row: pass 1 the row's largest magnitude, pass 2 eight elements per lane -> one dword of codes
__global__ __launch_bounds__(256) void mx4_quant_rows_kernel(const op16_t* __restrict__ W, int ldw, int N, int K, uint8_t* __restrict__ W4,
                                                             uint8_t* __restrict__ wexp) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const op16_t* row = W + (size_t)n * ldw;
  float amax = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const u32x4 v = *(const u32x4*)(row + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(oplo_to_f32(v[j])), fabsf(ophi_to_f32(v[j]))));
  }
  amax = wave_max(amax);
  const int e = mx4_exponent(amax);
  const float sc = mx4_scale(e);
  if (lane == 0) wexp[n] = (uint8_t)(e + 127);
  uint32_t* out = (uint32_t*)(W4 + (size_t)n * (K >> 1));
  for (int k = lane * 8; k < K; k += 512) {
    const u32x4 v = *(const u32x4*)(row + k);
    out[k >> 3] = mx4_pack8(oplo_to_f32(v[0]), ophi_to_f32(v[0]), oplo_to_f32(v[1]), ophi_to_f32(v[1]), oplo_to_f32(v[2]), ophi_to_f32(v[2]),
                            oplo_to_f32(v[3]), ophi_to_f32(v[3]), sc);
  }
}

// 16-bit lo [M][K] -> A4 + block exponents: a lane takes 8 consecutive k, a quad one 32-block
__global__ __launch_bounds__(256) void mx4_quant_lo_kernel(const op16_t* __restrict__ lo, int ld, int M, int K, uint8_t* __restrict__ A4,
                                                           uint8_t* __restrict__ aexp) {
  const int per_row = K >> 3;                                   // lanes per row
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int m = (int)(t / per_row), c = (int)(t % per_row);
  if (m >= M) return;                                           // (whole quads leave together: per_row % 4 == 0)
  const u32x4 v = *(const u32x4*)(lo + (size_t)m * ld + c * 8);
  float x[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    x[2 * j] = oplo_to_f32(v[j]);
    x[2 * j + 1] = ophi_to_f32(v[j]);
  }
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(x[j]));
  const int e = mx4_exponent(quad_max(amax));
  *(uint32_t*)(A4 + (size_t)m * (K >> 1) + c * 4) = mx4_pack8(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], mx4_scale(e));
  if ((c & 3) == 0) aexp[mx4_aexp_index(m, c >> 2, K >> 5)] = (uint8_t)(e + 127);
}

// RMSNorm (transformers-4.31 LlamaRMSNorm) with the level 3 operand out: hi = op16(v), lo = v - f32(hi) as MX-FP4.
// One wave per row; a lane holds EIGHT consecutive columns per chunk (two 16-byte loads, one 16-byte store of the 16-bit operand, one 4-byte
// store of codes: half as many store instructions as 4 columns per lane would issue, and a 32-block is one quad). The four waves of a
// workgroup take the FOUR rows whose block exponents share a dword of the scale array -- rows i, i + 16, i + 32, i + 48 of a 64-row group --
// park their exponent bytes in LDS and store them as whole dwords: a byte store per row and block, each to a line of its own, doubled this
// kernel's time (41 us against vt_rmsnorm's 19 at 5120 x 4096), and a wave walking the four rows one after the other tripled it (63 us).
template <int NCH>    // chunks of 512 columns
__global__ __launch_bounds__(256) void rmsnorm_mx_kernel(const float* __restrict__ x, const int* __restrict__ idx, const float* __restrict__ w,
                                                         op16_t* __restrict__ y, uint8_t* __restrict__ A4, uint8_t* __restrict__ aexp, int rows,
                                                         int D, float eps) {
  __shared__ __attribute__((aligned(4))) uint8_t se[NCH * 16 * 4];     // [block][row of the four]
  const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int rg = blockIdx.x >> 4, i16 = blockIdx.x & 15;
  const int row = rg * 64 + j * 16 + i16;
  const int KB = D >> 5;
  if (row < rows) {
    const float* xr = x + (size_t)(idx ? idx[row] : row) * D;
    f32x4 v[NCH][2];
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (lane + i * 64) * 8;
      if (c < D) {
        v[i][0] = *(const f32x4*)(xr + c);
        v[i][1] = *(const f32x4*)(xr + c + 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) sq += v[i][h][0] * v[i][h][0] + v[i][h][1] * v[i][h][1] + v[i][h][2] * v[i][h][2] + v[i][h][3] * v[i][h][3];
      } else {
        v[i][0] = v[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    op16_t* yr = y + (size_t)row * D;
    uint8_t* ar = A4 + (size_t)row * (D >> 1);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (lane + i * 64) * 8;
      if (c < D) {                                                // (D % 512 == 0: whole quads are in or out together)
        float val[8], lo[8];
        u32x4 o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 g = *(const f32x4*)(w + c + 4 * h);
#pragma unroll
          for (int r = 0; r < 4; ++r) val[4 * h + r] = v[i][h][r] * rstd * g[r];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[q] = pack_op2(val[2 * q], val[2 * q + 1]);
          lo[2 * q] = val[2 * q] - oplo_to_f32(o[q]);
          lo[2 * q + 1] = val[2 * q + 1] - ophi_to_f32(o[q]);
        }
        *(u32x4*)(yr + c) = o;
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) amax = fmaxf(amax, fabsf(lo[r]));
        const int e = mx4_exponent(quad_max(amax));
        *(uint32_t*)(ar + (c >> 1)) = mx4_pack8(lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7], mx4_scale(e));
        if ((lane & 3) == 0) se[(c >> 5) * 4 + j] = (uint8_t)(e + 127);
      }
    }
  } else {
    for (int kb = lane; kb < KB; kb += 64) se[kb * 4 + j] = 0;
  }
  __syncthreads();
  if ((int)threadIdx.x < KB) *(uint32_t*)(aexp + ((size_t)rg * KB + threadIdx.x) * 64 + i16 * 4) = *(const uint32_t*)(se + threadIdx.x * 4);
}

// LayerNorm (nn.LayerNorm: the towers' layer_norm2 / temporal_layer_norm2, reference modeling_video.py:72,82) with the level 3 operand out --
// the same organisation as rmsnorm_mx_kernel (one row per wave, the four rows of an exponent dword in one workgroup).
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_mx_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           op16_t* __restrict__ y, uint8_t* __restrict__ A4, uint8_t* __restrict__ aexp, int rows,
                                                           int D, float eps) {
  __shared__ __attribute__((aligned(4))) uint8_t se[NCH * 8 * 4];
  const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int rg = blockIdx.x >> 4, i16 = blockIdx.x & 15;
  const int row = rg * 64 + j * 16 + i16;
  const int KB = D >> 5;
  if (row < rows) {
    const float* xr = x + (size_t)row * D;
    f32x4 v[NCH];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (lane + i * 64) * 4;
      v[i] = (c < D) ? *(const f32x4*)(xr + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
      sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (lane + i * 64) * 4;
      if (c < D) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = v[i][r] - mean;
          sq += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    op16_t* yr = y + (size_t)row * D;
    uint8_t* ar = A4 + (size_t)row * (D >> 1);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (lane + i * 64) * 4;
      if (c < D) {
        const f32x4 g = *(const f32x4*)(gamma + c);
        const f32x4 b = *(const f32x4*)(beta + c);
        float val[4], lo[4];
        u32x2 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) val[r] = (v[i][r] - mean) * rstd * g[r] + b[r];
        o.x = pack_op2(val[0], val[1]);
        o.y = pack_op2(val[2], val[3]);
        *(u32x2*)(yr + c) = o;
        lo[0] = val[0] - oplo_to_f32(o.x);
        lo[1] = val[1] - ophi_to_f32(o.x);
        lo[2] = val[2] - oplo_to_f32(o.y);
        lo[3] = val[3] - ophi_to_f32(o.y);
        float amax = fmaxf(fmaxf(fabsf(lo[0]), fabsf(lo[1])), fmaxf(fabsf(lo[2]), fabsf(lo[3])));
        amax = quad_max(amax);
        amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
        const int e = mx4_exponent(amax);
        *(uint16_t*)(ar + (c >> 1)) = (uint16_t)mx4_pack4(lo[0], lo[1], lo[2], lo[3], mx4_scale(e));
        if ((lane & 7) == 0) se[(c >> 5) * 4 + j] = (uint8_t)(e + 127);
      }
    }
  } else {
    for (int kb = lane; kb < KB; kb += 64) se[kb * 4 + j] = 0;
  }
  __syncthreads();
  if ((int)threadIdx.x < KB) *(uint32_t*)(aexp + ((size_t)rg * KB + threadIdx.x) * 64 + i16 * 4) = *(const uint32_t*)(se + threadIdx.x * 4);
}

}  // namespace

int vt_layernorm_mx_launch(const float* x, const float* gamma, const float* beta, bf16_t* y, uint8_t* A4, uint8_t* aexp, int rows, int D,
                           float eps, hipStream_t s) {
  VT_REQUIRE(x && gamma && beta && y && A4 && aexp && rows > 0, "vt_layernorm_mx: null pointer");
  VT_REQUIRE(D % 256 == 0 && D <= 4096, "vt_layernorm_mx: D = %d must be a multiple of 256, at most 4096", D);
  const dim3 grid(cdiv(rows, 64) * 16), block(256);
  if (D <= 1024) hipLaunchKernelGGL(layernorm_mx_kernel<4>, grid, block, 0, s, x, gamma, beta, y, A4, aexp, rows, D, eps);
  else hipLaunchKernelGGL(layernorm_mx_kernel<16>, grid, block, 0, s, x, gamma, beta, y, A4, aexp, rows, D, eps);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_mx4_quant_weights_launch(const bf16_t* W, int ldw, int N, int K, uint8_t* W4, uint8_t* wexp, hipStream_t s) {
  VT_REQUIRE(W && W4 && wexp && N > 0 && K > 0 && (K % 8) == 0 && (ldw % 8) == 0, "vt_mx4_quant_weights: K and ldw must be multiples of 8");
  hipLaunchKernelGGL(mx4_quant_rows_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, W, ldw, N, K, W4, wexp);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_mx4_quant_lo_launch(const bf16_t* lo, int ld, int M, int K, uint8_t* A4, uint8_t* aexp, hipStream_t s) {
  VT_REQUIRE(lo && A4 && aexp && M > 0 && K > 0 && (K % 32) == 0 && (ld % 8) == 0, "vt_mx4_quant_lo: K must be a multiple of 32");
  const size_t threads = (size_t)M * (K >> 3);
  hipLaunchKernelGGL(mx4_quant_lo_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, lo, ld, M, K, A4, aexp);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_rmsnorm_mx_launch(const float* x, const int* idx, const float* w, bf16_t* y, uint8_t* A4, uint8_t* aexp, int rows, int D, float eps,
                         hipStream_t s) {
  VT_REQUIRE(x && w && y && A4 && aexp && rows > 0, "vt_rmsnorm_mx: null pointer");
  VT_REQUIRE(D % 512 == 0 && D <= 8192, "vt_rmsnorm_mx: D = %d must be a multiple of 512, at most 8192", D);
  const dim3 grid(cdiv(rows, 64) * 16), block(256);   // a workgroup per (64-row group, row % 16): its four waves take the four rows
  if (D <= 1024) hipLaunchKernelGGL(rmsnorm_mx_kernel<2>, grid, block, 0, s, x, idx, w, y, A4, aexp, rows, D, eps);
  else if (D <= 4096) hipLaunchKernelGGL(rmsnorm_mx_kernel<8>, grid, block, 0, s, x, idx, w, y, A4, aexp, rows, D, eps);
  else hipLaunchKernelGGL(rmsnorm_mx_kernel<16>, grid, block, 0, s, x, idx, w, y, A4, aexp, rows, D, eps);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

size_t vt_mx4_aexp_bytes(int M, int K) { return (size_t)cdiv(M, 256) * 256 * (size_t)(K >> 5); }
