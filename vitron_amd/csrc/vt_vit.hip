// vt_vit.hip -- LanguageBind ViT front end: patch gather (im2col) and embedding assembly + pre-LayerNorm.
//
// Replaces transformers-4.31 CLIPVisionEmbeddings (Conv2d k=s=P no bias -> flatten -> cat CLS -> +pos;
// SURVEY.md Appendix A) as used by the reference at
// vitron/model/multimodal_encoder/languagebind/video/modeling_video.py:602,637-650 (video: 'b c t h w ->
// (b t) c h w' rearrange, then embeddings, PatchDropout(identity at eval), pre_layrnorm).
// The convolution is a GEMM [F*G*G, 3*P*P] x [D, 3*P*P]^T with K index order (c, py, px); K is padded
// to a multiple of 64 for the MFMA tile kernel. The gather reads each patch row as one contiguous
// P-pixel segment straight out of NCHW / NCTHW -- no NHWC copy is made.
#include "vt_common.h"
#include "vt_kernels.h"

namespace {

template <typename PIX>
__device__ __forceinline__ float pix_to_f32(PIX v);
template <>
__device__ __forceinline__ float pix_to_f32<bf16_t>(bf16_t v) { return op_to_f32(v); }
template <>
__device__ __forceinline__ float pix_to_f32<float>(float v) { return v; }

// one thread per (patch, c, py): copies P pixels; a trailing group of threads zero-fills the K padding
template <typename PIX>
__global__ __launch_bounds__(256) void im2col_kernel(const PIX* __restrict__ pix, bf16_t* __restrict__ out, int B,
                                                     int T, int H, int W, int P, int Kpad, int video_layout) {
  const int G = H / P, Gw = W / P;
  const int segs = 3 * P;                    // (c, py) segments per patch
  const int pad = Kpad - 3 * P * P;
  const int units = segs + (pad > 0 ? 1 : 0);
  const long total = (long)B * T * G * Gw * units;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int u = (int)(i % units);
  const long patch = i / units;
  bf16_t* orow = out + patch * Kpad;
  if (u == segs) {
    for (int k = 3 * P * P; k < Kpad; ++k) orow[k] = 0;
    return;
  }
  const int c = u / P, py = u % P;
  const int gx = (int)(patch % Gw);
  const int gy = (int)((patch / Gw) % G);
  const int f = (int)(patch / ((long)G * Gw));
  const int b = f / T, t = f % T;
  size_t src;
  if (video_layout) src = ((((size_t)b * 3 + c) * T + t) * H + (gy * P + py)) * W + gx * P;   // [B][3][T][H][W]
  else src = (((size_t)f * 3 + c) * H + (gy * P + py)) * W + gx * P;                          // [F][3][H][W]
  bf16_t* o = orow + (c * P + py) * P;
  for (int px = 0; px < P; ++px) o[px] = f32_to_op(pix_to_f32<PIX>(pix[src + px]));
}

// x[f][0] = cls + pos[0]; x[f][1+p] = patch_out[f*G2+p] + pos[1+p]; x = LayerNorm(x) (fp32 in, fp32 out)
template <int NCH>
__global__ __launch_bounds__(256) void vit_embed_kernel(const float* __restrict__ patch_out,
                                                        const float* __restrict__ cls, const float* __restrict__ pos,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ x, int F, int G2, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int N = G2 + 1;
  if (row >= F * N) return;
  const int f = row / N, n = row % N;
  const float* src = (n == 0) ? cls : patch_out + ((size_t)f * G2 + (n - 1)) * D;
  const float* pr = pos + (size_t)n * D;
  f32x4 v[NCH];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < D) {
      v[i] = *(const f32x4*)(src + c) + *(const f32x4*)(pr + c);
      sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    } else {
      v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
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
  float* xr = x + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < D) {
      const f32x4 g = *(const f32x4*)(gamma + c);
      const f32x4 b = *(const f32x4*)(beta + c);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (v[i][r] - mean) * rstd * g[r] + b[r];
      *(f32x4*)(xr + c) = o;
    }
  }
}

__global__ void vit_attn_meta_kernel(int* __restrict__ seq_desc, int* __restrict__ tile_table, int F, int N) {
  const int ntiles = (N + 63) >> 6;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < F) {
    seq_desc[4 * i + 0] = i * N;
    seq_desc[4 * i + 1] = N;
    seq_desc[4 * i + 2] = N;
    seq_desc[4 * i + 3] = i * ntiles;
  }
  if (i < F * ntiles) tile_table[i] = i;
}

}  // namespace

int vt_vit_attn_meta_launch(int* seq_desc, int* tile_table, int F, int N, hipStream_t s) {
  const int n = F * ((N + 63) / 64);
  hipLaunchKernelGGL(vit_attn_meta_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, seq_desc, tile_table, F, N);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_im2col_launch(const void* pixels, int pix_dtype, bf16_t* patches, int B, int T, int H, int W, int P,
                     int Kpad, int video_layout, hipStream_t s) {
  VT_REQUIRE(pixels && patches, "vt_im2col: null pointer");
  VT_REQUIRE(B > 0 && T > 0 && P > 0 && H % P == 0 && W % P == 0, "vt_im2col: H=%d W=%d must be multiples of P=%d", H, W, P);
  VT_REQUIRE(Kpad >= 3 * P * P && Kpad % 8 == 0, "vt_im2col: Kpad=%d too small / not a multiple of 8", Kpad);
  const int units = 3 * P + (Kpad > 3 * P * P ? 1 : 0);
  const long total = (long)B * T * (H / P) * (W / P) * units;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (pix_dtype == VT_DTYPE_BF16)
    hipLaunchKernelGGL((im2col_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)pixels, patches, B, T, H, W, P, Kpad, video_layout);
  else if (pix_dtype == VT_DTYPE_F32)
    hipLaunchKernelGGL((im2col_kernel<float>), grid, block, 0, s, (const float*)pixels, patches, B, T, H, W, P, Kpad, video_layout);
  else {
    vt_set_error("vt_im2col: unsupported pixel dtype %d", pix_dtype);
    return VT_ERR_ARG;
  }
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_vit_embed_launch(const float* patch_out, const float* cls, const float* pos, const float* g, const float* b,
                        float* x, int F, int G2, int D, float eps, hipStream_t s) {
  VT_REQUIRE(patch_out && cls && pos && g && b && x, "vt_vit_embed: null pointer");
  VT_REQUIRE(D % 4 == 0 && D <= 4096, "vt_vit_embed: D=%d unsupported", D);
  const int rows = F * (G2 + 1);
  dim3 grid(cdiv(rows, 4)), block(256);
  const int nch = cdiv(D, 256);
  if (nch <= 1) hipLaunchKernelGGL((vit_embed_kernel<1>), grid, block, 0, s, patch_out, cls, pos, g, b, x, F, G2, D, eps);
  else if (nch <= 2) hipLaunchKernelGGL((vit_embed_kernel<2>), grid, block, 0, s, patch_out, cls, pos, g, b, x, F, G2, D, eps);
  else if (nch <= 4) hipLaunchKernelGGL((vit_embed_kernel<4>), grid, block, 0, s, patch_out, cls, pos, g, b, x, F, G2, D, eps);
  else if (nch <= 8) hipLaunchKernelGGL((vit_embed_kernel<8>), grid, block, 0, s, patch_out, cls, pos, g, b, x, F, G2, D, eps);
  else hipLaunchKernelGGL((vit_embed_kernel<16>), grid, block, 0, s, patch_out, cls, pos, g, b, x, F, G2, D, eps);
  VT_LAUNCH_CHECK();
  return VT_OK;
}
