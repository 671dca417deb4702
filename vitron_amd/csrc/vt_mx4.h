// vt_mx4.h -- device helpers shared by the producers of MX-FP4 operand images (vt_mx4.hip, the SwiGLU epilogue of vt_gemm8x.inc, the
// attention epilogues): block exponent, e2m1 conversion, and where a block's exponent byte lives. The format, its rounding rule and the
// scale layout are described in vt_mx4.hip / include/vitron_hip.h; oracle/vitron_oracle.py mx4_quant restates them.
#pragma once
#include "vt_common.h"

// unbiased scale exponent of a block with largest magnitude amax: the smallest e with amax / 2^e <= 6 -- floor(log2 amax) - 2, +1 when the
// mantissa exceeds 1.5 -- clamped to [-126, 127] (so that 2^e is a normal fp32 number; a block of zeros gets -126)
__device__ __forceinline__ int mx4_exponent(float amax) {
  const uint32_t b = __float_as_uint(amax);
  const int e = (int)((b >> 23) & 0xffu) - 129 + ((b & 0x7fffffu) > 0x400000u ? 1 : 0);
  return max(-126, min(127, e));
}
__device__ __forceinline__ float mx4_scale(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }
// eight values -> eight e2m1 codes (v[0] in bits 3:0): v_cvt_scalef32_pk_fp4_f32 = code(x / scale), round to nearest, ties to the even
// mantissa, saturating at +-6 (tools/mx_probe.hip compares it with the arithmetic restatement value by value)
__device__ __forceinline__ uint32_t mx4_pack8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, float scale) {
  uint32_t o = 0;
  o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, v0, v1, scale, 0);
  o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, v2, v3, scale, 1);
  o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, v4, v5, scale, 2);
  o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, v6, v7, scale, 3);
  return o;
}
__device__ __forceinline__ uint32_t mx4_pack4(float v0, float v1, float v2, float v3, float scale) {   // four codes in bits 15:0
  uint32_t o = 0;
  o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, v0, v1, scale, 0);
  o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, v2, v3, scale, 1);
  return o;
}
// byte index of the exponent of (row m, 32-block kb) in the order the GEMM's lanes fetch them: [m / 64][kb][m % 16][(m % 64) / 16]
__device__ __forceinline__ size_t mx4_aexp_index(int m, int kb, int KB) {
  return ((size_t)(m >> 6) * KB + kb) * 64 + (m & 15) * 4 + ((m & 63) >> 4);
}

// Attention epilogues (vt_attn.hip, vt_attn_w4.hip): a lane holds 16 values of ONE 32-block of its output row -- d = 8 g + 4 hh + j within
// the block, g = 0..3, j = 0..3, hh = lane >> 5; the other 16 sit in lane ^ 32. lo[4 g + j] = the remainders v - f32(op16(v)). Writes the
// block's 16 code bytes (each lane 8 of them: the two lanes exchange halves so that a lane stores whole dwords) and its exponent byte.
// Every lane of the wave must call it (shuffles); `valid` masks the stores.
__device__ __forceinline__ void mx4_store_attn_block(const float (&lo)[16], int hh, bool valid, uint8_t* block16, uint8_t* ebyte) {
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(lo[i]));
  amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
  const int e = mx4_exponent(amax);
  const float sc = mx4_scale(e);
  const uint32_t m01 = mx4_pack4(lo[0], lo[1], lo[2], lo[3], sc) | (mx4_pack4(lo[4], lo[5], lo[6], lo[7], sc) << 16);
  const uint32_t m23 = mx4_pack4(lo[8], lo[9], lo[10], lo[11], sc) | (mx4_pack4(lo[12], lo[13], lo[14], lo[15], sc) << 16);
  const uint32_t o01 = __shfl_xor(m01, 32, 64), o23 = __shfl_xor(m23, 32, 64);
  u32x2 d;   // dword of g = codes of (g, hh = 0) | codes of (g, hh = 1) << 16; lane hh = 0 stores g = 0, 1, lane hh = 1 stores g = 2, 3
  if (hh == 0) {
    d.x = (m01 & 0xffffu) | (o01 << 16);
    d.y = (m01 >> 16) | (o01 & 0xffff0000u);
  } else {
    d.x = (o23 & 0xffffu) | (m23 << 16);
    d.y = (o23 >> 16) | (m23 & 0xffff0000u);
  }
  if (valid) {
    *(u32x2*)(block16 + hh * 8) = d;
    if (hh == 0) *ebyte = (uint8_t)(e + 127);
  }
}
