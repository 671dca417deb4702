// vt_region.hip -- region_extractor front half: box -> patch-grid cell mask -> masked mean of patch features.
//
// Replaces RegionExtractor.transform_bbox_2_mask + MaskPooling of the reference
// (vitron/model/region_extractor/layer.py:23-43 and :77-85, called at :112-120):
//   mask[int(x1):int(x2), int(y1):int(y2)] = 1   on an image_size^2 canvas   (x indexes ROWS, :83)
//   mask = bilinear_resize(mask, GxG, align_corners=False) > 0                (:31,35)
//   pooled[c] = sum_cells feats[c, cell] * mask[cell] / (sum(mask) + 1e-8)     (:36-42)
// The bilinear weights of a cell are non-negative and the canvas is {0,1}, so "> 0" holds iff one of the
// (at most 2x2) source taps with non-zero weight lies inside the box -- an integer test. The box arrives as
// already-resolved Python slice bounds [r0, r1) x [c0, c1) (host does int() and slice clamping exactly as
// the reference's Python does), so the cell mask is bit-exact by construction.
// The MLP / location-encoder half (layer.py:7-20,46-56,123-130) runs on the skinny GEMM kernel.
#include "vt_common.h"
#include "vt_kernels.h"

namespace {

// PyTorch upsample_bilinear2d source taps (align_corners=False): src = max(scale*(dst+0.5)-0.5, 0)
__device__ __forceinline__ void bilinear_taps(int dst, float scale, int in_size, int& i0, int& i1, float& w0, float& w1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = (i0 < in_size - 1) ? i0 + 1 : i0;
  w1 = src - (float)i0;
  w0 = 1.f - w1;
}

__global__ __launch_bounds__(256) void region_pool_kernel(const bf16_t* __restrict__ feats,
                                                          const int* __restrict__ slices, int G, int image_size,
                                                          int D, bf16_t* __restrict__ pooled,
                                                          int* __restrict__ cell_mask, int* __restrict__ cell_count) {
  extern __shared__ int s_mask[];  // G*G cells + 1 counter
  const int b = blockIdx.x;
  const int r0 = slices[b * 4 + 0], r1 = slices[b * 4 + 1], c0 = slices[b * 4 + 2], c1 = slices[b * 4 + 3];
  const int cells = G * G;
  const float scale = (float)image_size / (float)G;
  if (threadIdx.x == 0) s_mask[cells] = 0;
  __syncthreads();
  for (int cell = threadIdx.x; cell < cells; cell += blockDim.x) {
    const int i = cell / G, j = cell % G;
    int ia, ib, ja, jb;
    float wa, wb, va, vb;
    bilinear_taps(i, scale, image_size, ia, ib, wa, wb);
    bilinear_taps(j, scale, image_size, ja, jb, va, vb);
    const bool row_a = (wa > 0.f) && (ia >= r0 && ia < r1);
    const bool row_b = (wb > 0.f) && (ib >= r0 && ib < r1);
    const bool col_a = (va > 0.f) && (ja >= c0 && ja < c1);
    const bool col_b = (vb > 0.f) && (jb >= c0 && jb < c1);
    const int on = ((row_a || row_b) && (col_a || col_b)) ? 1 : 0;
    s_mask[cell] = on;
    if (cell_mask) cell_mask[(size_t)b * cells + cell] = on;
    if (on) atomicAdd(&s_mask[cells], 1);
  }
  __syncthreads();
  const int count = s_mask[cells];
  if (threadIdx.x == 0 && cell_count) cell_count[b] = count;
  const float inv = 1.0f / ((float)count + 1e-8f);
  // each thread owns 4 consecutive channels (D/4 threads active per pass)
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int cell = 0; cell < cells; ++cell) {
      if (!s_mask[cell]) continue;
      const u32x2 w = *(const u32x2*)(feats + ((size_t)b * cells + cell) * D + c);
      // reference: x * (mask / denorm) summed over cells; same value, summed in fp32
      a0 += bf16lo_to_f32(w.x) * inv;
      a1 += bf16hi_to_f32(w.x) * inv;
      a2 += bf16lo_to_f32(w.y) * inv;
      a3 += bf16hi_to_f32(w.y) * inv;
    }
    u32x2 o;
    o.x = pack_bf16x2(a0, a1);
    o.y = pack_bf16x2(a2, a3);
    *(u32x2*)(pooled + (size_t)b * D + c) = o;
  }
}

}  // namespace

int vt_region_pool_launch(const bf16_t* feats, const int* slices, int B, int G, int image_size, int D,
                          bf16_t* pooled, int* cell_mask, int* cell_count, hipStream_t s) {
  VT_REQUIRE(feats && slices && pooled, "vt_region_pool: null pointer");
  VT_REQUIRE(B > 0 && G > 0 && image_size >= G && D % 4 == 0, "vt_region_pool: bad shape B=%d G=%d image=%d D=%d", B, G, image_size, D);
  const int smem = (G * G + 1) * (int)sizeof(int);
  hipLaunchKernelGGL(region_pool_kernel, dim3(B), dim3(256), smem, s, feats, slices, G, image_size, D, pooled,
                     cell_mask, cell_count);
  VT_LAUNCH_CHECK();
  return VT_OK;
}
