// vt_llama.hip -- small kernels around the decoder: embedding gather + multimodal splice, greedy argmax.
//
// embed_splice replaces the tensor surgery of prepare_inputs_labels_for_multimodal in the reference
// (vitron/model/llava_arch.py:306-398 region branch, :479-558 plain branch): embed_tokens gather for text
// ids, visual feature blocks at every -200 sentinel, one region feature row at every -300 sentinel,
// truncation and zero padding. The integer plan (which source row feeds which output row) is computed on
// the host by vitron_amd.model.llava_arch.build_splice_plan, bit-exactly following the reference's list
// logic; this kernel only moves rows.
#include "vt_common.h"
#include "vt_kernels.h"

namespace {

// plan[r] = {kind, index}: kind 0 = token id into tok_table, 1 = row of vis, 2 = row of reg, 3 = zeros
__global__ __launch_bounds__(256) void embed_splice_kernel(const bf16_t* __restrict__ tok_table,
                                                           const bf16_t* __restrict__ vis,
                                                           const bf16_t* __restrict__ reg,
                                                           const int* __restrict__ plan, int rows, int H,
                                                           bf16_t* __restrict__ out) {
  const int chunks = H >> 3;  // 16-B chunks per row
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)rows * chunks;
       i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / chunks), c = (int)(i % chunks);
    const int kind = plan[2 * r], idx = plan[2 * r + 1];
    u32x4 v = {0u, 0u, 0u, 0u};
    if (kind == 0) v = *(const u32x4*)(tok_table + (size_t)idx * H + c * 8);
    else if (kind == 1) v = *(const u32x4*)(vis + (size_t)idx * H + c * 8);
    else if (kind == 2) v = *(const u32x4*)(reg + (size_t)idx * H + c * 8);
    *(u32x4*)(out + (size_t)r * H + c * 8) = v;
  }
}

// first index of the maximum of each row (torch.argmax tie rule on CPU: lowest index)
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int V, int ldl,
                                                     int* __restrict__ out) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const float* row = logits + (size_t)blockIdx.x * ldl;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    const float v = row[i];
    if (v > best || (v == best && i < bi)) {
      best = v;
      bi = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sv[wave] = best;
    si[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
        best = sv[w];
        bi = si[w];
      }
    out[blockIdx.x] = bi;
  }
}

}  // namespace

int vt_embed_splice_launch(const bf16_t* tok_table, const bf16_t* vis, const bf16_t* reg, const int* plan, int rows,
                           int H, bf16_t* out, hipStream_t s) {
  VT_REQUIRE(tok_table && plan && out, "vt_embed_splice: null pointer");
  VT_REQUIRE(rows > 0 && H % 8 == 0, "vt_embed_splice: rows=%d H=%d (H must be a multiple of 8)", rows, H);
  const long total = (long)rows * (H / 8);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(embed_splice_kernel, dim3(blocks), dim3(256), 0, s, tok_table, vis, reg, plan, rows, H, out);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_argmax_launch(const float* logits, int rows, int V, int ldl, int* out_ids, hipStream_t s) {
  VT_REQUIRE(logits && out_ids && rows > 0 && V > 0, "vt_argmax: bad arguments");
  hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, s, logits, V, ldl, out_ids);
  VT_LAUNCH_CHECK();
  return VT_OK;
}
