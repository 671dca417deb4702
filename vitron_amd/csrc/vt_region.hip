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

// One launch for everything in RegionExtractor.forward that is not a 4096-wide Linear:
//   grid (box, 64-channel slab), 4 waves. Wave 0 evaluates the row / column tests for the <= 64 grid lines with one ballot each
//   (the cell mask of a box is the product set rows x cols: a cell is on iff its row line AND its column line touch the box),
//   every wave then walks the bounding rectangle of the on-cells, 8 cells per load instruction (8 lanes x 16 bytes = the slab's
//   128 bytes of one cell), accumulating x * 1/(count + 1e-8) in fp32; the 8 cell-lanes are folded with three xor-shuffles and
//   the 4 waves through LDS. Slab 0 of every box also writes the int32 cell mask and the count (bit-exact outputs).
//   The blocks also evaluate their share of LocationEncoder's first layer, relu(coords . W0^T + b0) (K = 4: layer.py:46-56,126),
//   so that the 3-layer MLP and the location branch end in ONE weight-streaming GEMM over the concatenated K (vt_api.hip).
__global__ __launch_bounds__(256) void region_pool_kernel(const bf16_t* __restrict__ feats, const int* __restrict__ slices, int G,
                                                          int image_size, int D, bf16_t* __restrict__ pooled,
                                                          int* __restrict__ cell_mask, int* __restrict__ cell_count,
                                                          const float* __restrict__ coords, const bf16_t* __restrict__ loc_w0,
                                                          const float* __restrict__ loc_b0, int loc_n, bf16_t* __restrict__ loc_out,
                                                          int ld_loc) {
  __shared__ unsigned long long s_bits[2];
  __shared__ float s_part[4][64];
  const int b = blockIdx.x, slab = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cells = G * G;
  if (wave == 0) {
    const int r0 = slices[b * 4 + 0], r1 = slices[b * 4 + 1], c0 = slices[b * 4 + 2], c1 = slices[b * 4 + 3];
    const float scale = (float)image_size / (float)G;
    int ia, ib;
    float wa, wb;
    bilinear_taps(min(lane, G - 1), scale, image_size, ia, ib, wa, wb);   // the taps of grid line `lane` (rows and columns alike)
    const bool live = lane < G;
    const bool row_on = live && (((wa > 0.f) && ia >= r0 && ia < r1) || ((wb > 0.f) && ib >= r0 && ib < r1));
    const bool col_on = live && (((wa > 0.f) && ia >= c0 && ia < c1) || ((wb > 0.f) && ib >= c0 && ib < c1));
    const unsigned long long rm = __ballot(row_on), cm = __ballot(col_on);
    if (lane == 0) {
      s_bits[0] = rm;
      s_bits[1] = cm;
    }
  }
  __syncthreads();
  const unsigned long long rm = s_bits[0], cm = s_bits[1];
  const int count = __popcll(rm) * __popcll(cm);
  if (slab == 0) {
    if (cell_mask)
      for (int cell = threadIdx.x; cell < cells; cell += blockDim.x)
        cell_mask[(size_t)b * cells + cell] = (int)((rm >> (cell / G)) & 1ull) & (int)((cm >> (cell % G)) & 1ull);
    if (cell_count && threadIdx.x == 0) cell_count[b] = count;
  }
  // ---- masked mean over the bounding rectangle of the on-cells ------------------------------------------------------------
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  if (count > 0) {
    const int i0 = __ffsll((long long)rm) - 1, i1 = 63 - __clzll((long long)rm);
    const int j0 = __ffsll((long long)cm) - 1, j1 = 63 - __clzll((long long)cm);
    const int nj = j1 - j0 + 1, total = (i1 - i0 + 1) * nj;
    const float inv = 1.0f / ((float)count + 1e-8f);
    const bf16_t* base = feats + (size_t)b * cells * D + slab * 64 + (lane & 7) * 8;
#pragma unroll 4
    for (int k = wave * 8 + (lane >> 3); k < total; k += 32) {
      const int i = i0 + k / nj, j = j0 + k % nj;
      if (!(((rm >> i) & 1ull) && ((cm >> j) & 1ull))) continue;      // (the on-lines of a box are contiguous: never taken)
      const u32x4 w = *(const u32x4*)(base + (size_t)(i * G + j) * D);
      // reference: x * (mask / denorm) summed over the cells (layer.py:36-42); same products, summed in fp32
      acc[0] += oplo_to_f32(w.x) * inv;
      acc[1] += ophi_to_f32(w.x) * inv;
      acc[2] += oplo_to_f32(w.y) * inv;
      acc[3] += ophi_to_f32(w.y) * inv;
      acc[4] += oplo_to_f32(w.z) * inv;
      acc[5] += ophi_to_f32(w.z) * inv;
      acc[6] += oplo_to_f32(w.w) * inv;
      acc[7] += ophi_to_f32(w.w) * inv;
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float v = acc[c];
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    acc[c] = v;
  }
  if (lane < 8) {
#pragma unroll
    for (int c = 0; c < 8; ++c) s_part[wave][lane * 8 + c] = acc[c];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const float v = (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
    pooled[(size_t)b * D + slab * 64 + threadIdx.x] = f32_to_op(v);
  }
  // ---- LocationEncoder layer 0: this block's share of the loc_n outputs ------------------------------------------------------
  if (loc_out) {
    const int per = (loc_n + gridDim.y - 1) / gridDim.y;
    for (int t = threadIdx.x; t < per; t += blockDim.x) {
      const int n = slab * per + t;
      if (n >= loc_n) break;
      const f32x4 cw = *(const f32x4*)(coords + (size_t)b * 4);        // {x1,y1,x2,y2} in fp32: a K = 4 layer costs nothing in full
      const u32x2 ww = *(const u32x2*)(loc_w0 + (size_t)n * 8);         // precision, and integers above 256 are not bf16 values
      float v = cw[0] * oplo_to_f32(ww.x);
      v = fmaf(cw[1], ophi_to_f32(ww.x), v);
      v = fmaf(cw[2], oplo_to_f32(ww.y), v);
      v = fmaf(cw[3], ophi_to_f32(ww.y), v);
      v += loc_b0[n];
      loc_out[(size_t)b * ld_loc + n] = f32_to_op(fmaxf(v, 0.f));
    }
  }
}

}  // namespace

int vt_region_pool_launch(const bf16_t* feats, const int* slices, int B, int G, int image_size, int D,
                          bf16_t* pooled, int* cell_mask, int* cell_count, const float* coords, const bf16_t* loc_w0,
                          const float* loc_b0, int loc_n, bf16_t* loc_out, int ld_loc, hipStream_t s) {
  VT_REQUIRE(feats && slices && pooled, "vt_region_pool: null pointer");
  VT_REQUIRE(B > 0 && G > 0 && G <= 64 && image_size >= G && D % 64 == 0,
             "vt_region_pool: bad shape B=%d G=%d (<= 64) image=%d D=%d (multiple of 64)", B, G, image_size, D);
  if (loc_out) VT_REQUIRE(coords && loc_w0 && loc_b0 && loc_n > 0, "vt_region_pool: location-encoder inputs missing");
  hipLaunchKernelGGL(region_pool_kernel, dim3(B, D / 64), dim3(256), 0, s, feats, slices, G, image_size, D, pooled, cell_mask,
                     cell_count, coords, loc_w0, loc_b0, loc_n, loc_out, ld_loc);
  VT_LAUNCH_CHECK();
  return VT_OK;
}
