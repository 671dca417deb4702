// vt_preproc.hip -- GPU pre-processing of decoded frames: resize (short side) + centre crop + scale + normalise, fused.
//
// The step immediately in front of the towers. Restates, for tensors already on the device:
//   image: ToTensor -> Resize(S, BICUBIC) -> CenterCrop(S) -> Normalize(mean, std)
//          (reference .../languagebind/image/processing_image.py:15-25; torchvision's tensor Resize = F.interpolate(
//           mode='bicubic', align_corners=False, antialias=False), cubic coefficient A = -0.75, border taps clamped)
//   video: x/255 -> NormalizeVideo -> ShortSideScale(S) -> CenterCropVideo(S) [-> RandomHorizontalFlipVideo]
//          (reference .../languagebind/video/processing_video.py:45-53; pytorchvideo short_side_scale = F.interpolate(
//           mode='bilinear', align_corners=False), new long side = floor(long/short * S))
// Normalisation is affine and the interpolation weights sum to 1, so normalising after the resample (one pass, reading
// the uint8 source once) equals the reference's order up to fp32 rounding. The flip is a parameter (the reference applies
// it RANDOMLY at inference, p=0.5); default off.
// One thread per output pixel, all three channels; source is HWC (what PIL / decord / OpenCV hand over) or CHW.
#include "vt_common.h"
#include "vt_kernels.h"

namespace {

struct PreP {
  const void* src;
  void* dst;
  int F, H, W;            // source frames
  int RH, RW;             // resized size (virtual; never materialised)
  int top, left, S;       // crop window inside the resized image, output S x S
  float scale_h, scale_w; // source / resized
  float pre;              // multiplied into the source value (1/255 for uint8)
  float mean[3], istd[3];
  long dst_sc, dst_sf;    // output strides (elements) of channel and frame
  int flip;
};

template <typename T>
__device__ __forceinline__ float ld(const T* p, long i);
template <>
__device__ __forceinline__ float ld<uint8_t>(const uint8_t* p, long i) { return (float)p[i]; }
template <>
__device__ __forceinline__ float ld<float>(const float* p, long i) { return p[i]; }

__device__ __forceinline__ void cubic_w(float t, float* w) {  // PyTorch get_cubic_upsample_coefficients, A = -0.75
  const float A = -0.75f;
  float x = t + 1.0f;
  w[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
  x = t;
  w[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
  x = 1.0f - t;
  w[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
  x = 2.0f - t;
  w[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
}

template <typename SRC, typename DST, bool HWC, bool BICUBIC>
__global__ __launch_bounds__(256) void preprocess_kernel(PreP p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)p.F * p.S * p.S;
  if (idx >= total) return;
  const int x = (int)(idx % p.S), y = (int)((idx / p.S) % p.S), f = (int)(idx / ((long)p.S * p.S));
  const int ox = p.flip ? (p.S - 1 - x) : x;
  const int ry = y + p.top, rx = ox + p.left;      // pixel of the (virtual) resized image
  const SRC* s = (const SRC*)p.src;
  const long fbase = (long)f * p.H * p.W * 3;
  auto at = [&](int yy, int xx, int c) -> float {
    yy = min(max(yy, 0), p.H - 1);
    xx = min(max(xx, 0), p.W - 1);
    return HWC ? ld<SRC>(s, fbase + ((long)yy * p.W + xx) * 3 + c) : ld<SRC>(s, fbase + ((long)c * p.H + yy) * p.W + xx);
  };
  float v[3] = {0.f, 0.f, 0.f};
  if (BICUBIC) {  // area_pixel_compute_source_index(align_corners=False, cubic=True): no clamp of the negative index
    const float sy = p.scale_h * ((float)ry + 0.5f) - 0.5f, sx = p.scale_w * ((float)rx + 0.5f) - 0.5f;
    const int iy = (int)floorf(sy), ix = (int)floorf(sx);
    float wy[4], wx[4];
    cubic_w(sy - (float)iy, wy);
    cubic_w(sx - (float)ix, wx);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) r += at(iy - 1 + j, ix - 1 + i, c) * wx[i];
        acc += r * wy[j];
      }
      v[c] = acc;
    }
  } else {  // bilinear, align_corners=False: negative source index clamped to 0
    float sy = p.scale_h * ((float)ry + 0.5f) - 0.5f, sx = p.scale_w * ((float)rx + 0.5f) - 0.5f;
    sy = fmaxf(sy, 0.f);
    sx = fmaxf(sx, 0.f);
    const int y0 = min((int)sy, p.H - 1), x0 = min((int)sx, p.W - 1);
    const int y1 = min(y0 + 1, p.H - 1), x1 = min(x0 + 1, p.W - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      v[c] = (1.f - ly) * ((1.f - lx) * at(y0, x0, c) + lx * at(y0, x1, c)) + ly * ((1.f - lx) * at(y1, x0, c) + lx * at(y1, x1, c));
  }
  DST* d = (DST*)p.dst;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float o = (v[c] * p.pre - p.mean[c]) * p.istd[c];
    const long di = (long)c * p.dst_sc + (long)f * p.dst_sf + (long)y * p.S + x;
    if constexpr (sizeof(DST) == 2) d[di] = f32_to_op(o);
    else d[di] = o;
  }
}

}  // namespace

int vt_preprocess_launch(const void* src, int src_u8, int hwc, int F, int H, int W, int bicubic, int S, const float* mean,
                         const float* std, int flip, void* dst, int dst_dtype, long dst_sc, long dst_sf, hipStream_t s) {
  VT_REQUIRE(src && dst && mean && std, "vt_preprocess: null pointer");
  VT_REQUIRE(F > 0 && H > 0 && W > 0 && S > 0, "vt_preprocess: bad shape");
  PreP p;
  p.src = src; p.dst = dst; p.F = F; p.H = H; p.W = W; p.S = S; p.flip = flip;
  // short side -> S, long side: torchvision Resize(int) uses int(S * long / short); pytorchvideo floor((long/short) * S)
  if (W <= H) {
    p.RW = S;
    p.RH = bicubic ? (int)((long)S * H / W) : (int)floor(((double)H / (double)W) * S);
  } else {
    p.RH = S;
    p.RW = bicubic ? (int)((long)S * W / H) : (int)floor(((double)W / (double)H) * S);
  }
  p.scale_h = (float)H / (float)p.RH;
  p.scale_w = (float)W / (float)p.RW;
  p.top = (int)lrint((p.RH - S) / 2.0);
  p.left = (int)lrint((p.RW - S) / 2.0);
  p.pre = src_u8 ? (1.0f / 255.0f) : 1.0f;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean[c]; p.istd[c] = 1.0f / std[c]; }
  p.dst_sc = dst_sc; p.dst_sf = dst_sf;
  const long total = (long)F * S * S;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define VT_PRE(SRC, DST, HW, BC) hipLaunchKernelGGL((preprocess_kernel<SRC, DST, HW, BC>), grid, block, 0, s, p)
#define VT_PRE2(SRC, DST)                                                   \
  do {                                                                      \
    if (hwc) { if (bicubic) VT_PRE(SRC, DST, true, true); else VT_PRE(SRC, DST, true, false); } \
    else { if (bicubic) VT_PRE(SRC, DST, false, true); else VT_PRE(SRC, DST, false, false); }   \
  } while (0)
  if (src_u8) { if (dst_dtype == VT_DTYPE_BF16) VT_PRE2(uint8_t, bf16_t); else VT_PRE2(uint8_t, float); }
  else { if (dst_dtype == VT_DTYPE_BF16) VT_PRE2(float, bf16_t); else VT_PRE2(float, float); }
#undef VT_PRE2
#undef VT_PRE
  VT_LAUNCH_CHECK();
  return VT_OK;
}
