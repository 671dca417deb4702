// vt_norm.hip -- row normalisations and casts. All are HBM-bound: one wavefront per row, 16-byte
// loads, statistics in fp32, a single pass over the row held in registers.
//
// The residual stream of both towers is kept in fp32 (the MFMA GEMM epilogue accumulates into it), so
// the norms read fp32 and emit the bf16 operand of the next GEMM.
//   LayerNorm : nn.LayerNorm(1024, eps) x4 per ViT layer -- reference
//               vitron/model/multimodal_encoder/languagebind/video/modeling_video.py:70,72,82,604
//               with the temporal-embedding add of :110-114 fused in front of temporal_layer_norm1.
//   RMSNorm   : transformers-4.31 LlamaRMSNorm, used by the decoder the
//               reference drives at vitron/model/language_model/llava_llama.py:91-102.
#include "vt_common.h"
#include "vt_kernels.h"

namespace {

template <int NCH>  // float4 chunks per lane; row fits when D <= NCH*256
__global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, const float* __restrict__ temb,
                                                        int T, int tokens_per_frame,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                        int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* xr = x + (size_t)row * D;
  const float* te = nullptr;
  if (temb) te = temb + (size_t)((row / tokens_per_frame) % T) * D;
  f32x4 v[NCH];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < D) {
      v[i] = *(const f32x4*)(xr + c);
      if (te) {
        v[i] += *(const f32x4*)(te + c);
        *(f32x4*)(xr + c) = v[i];
      }
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
  bf16_t* yr = y + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < D) {
      const f32x4 g = *(const f32x4*)(gamma + c);
      const f32x4 b = *(const f32x4*)(beta + c);
      u32x2 o;
      o.x = pack_op2((v[i][0] - mean) * rstd * g[0] + b[0], (v[i][1] - mean) * rstd * g[1] + b[1]);
      o.y = pack_op2((v[i][2] - mean) * rstd * g[2] + b[2], (v[i][3] - mean) * rstd * g[3] + b[3]);
      *(u32x2*)(yr + c) = o;
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                      const float* __restrict__ w, bf16_t* __restrict__ y, int rows,
                                                      int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)(idx ? idx[row] : row) * D;
  f32x4 v[NCH];
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < D) {
      v[i] = *(const f32x4*)(xr + c);
      sq += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    } else {
      v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
  bf16_t* yr = y + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < D) {
      const f32x4 g = *(const f32x4*)(w + c);
      u32x2 o;
      o.x = pack_op2(v[i][0] * rstd * g[0], v[i][1] * rstd * g[1]);
      o.y = pack_op2(v[i][2] * rstd * g[2], v[i][3] * rstd * g[3]);
      *(u32x2*)(yr + c) = o;
    }
  }
}

// precise_qk mode (vt_llama_model.precise_qk): the norm output as an operand PAIR -- y = op(x_n) and y_lo = op(x_n - f32(y)) -- so that
// the q / k projections can run as A_hi.W^T + A_lo.W^T (16 + 16 mantissa bits of the A operand instead of 11 / 8)
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_hilo_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                           const float* __restrict__ w, bf16_t* __restrict__ y,
                                                           bf16_t* __restrict__ ylo, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)(idx ? idx[row] : row) * D;
  f32x4 v[NCH];
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < D) {
      v[i] = *(const f32x4*)(xr + c);
      sq += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    } else {
      v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < D) {
      const f32x4 g = *(const f32x4*)(w + c);
      const float a0 = v[i][0] * rstd * g[0], a1 = v[i][1] * rstd * g[1], a2 = v[i][2] * rstd * g[2], a3 = v[i][3] * rstd * g[3];
      u32x2 hi, lo;
      hi.x = pack_op2(a0, a1);
      hi.y = pack_op2(a2, a3);
      lo.x = pack_op2(a0 - oplo_to_f32(hi.x), a1 - ophi_to_f32(hi.x));
      lo.y = pack_op2(a2 - oplo_to_f32(hi.y), a3 - ophi_to_f32(hi.y));
      *(u32x2*)(y + (size_t)row * D + c) = hi;
      *(u32x2*)(ylo + (size_t)row * D + c) = lo;
    }
  }
}

// ---- precise (pair-operand) mode, level 2: the remaining producers of GEMM A operands as pairs (hi = op(v), lo = op(v - f32(hi))) ----
__device__ __forceinline__ void store_pair4(bf16_t* hi_p, bf16_t* lo_p, float a0, float a1, float a2, float a3) {
  u32x2 hi, lo;
  hi.x = pack_op2(a0, a1);
  hi.y = pack_op2(a2, a3);
  lo.x = pack_op2(a0 - oplo_to_f32(hi.x), a1 - ophi_to_f32(hi.x));
  lo.y = pack_op2(a2 - oplo_to_f32(hi.y), a3 - ophi_to_f32(hi.y));
  *(u32x2*)hi_p = hi;
  *(u32x2*)lo_p = lo;
}

template <int NCH>
__global__ __launch_bounds__(256) void layernorm_hilo_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                             bf16_t* __restrict__ ylo, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
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
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < D) {
      const f32x4 g = *(const f32x4*)(gamma + c);
      const f32x4 b = *(const f32x4*)(beta + c);
      store_pair4(y + (size_t)row * D + c, ylo + (size_t)row * D + c, (v[i][0] - mean) * rstd * g[0] + b[0],
                  (v[i][1] - mean) * rstd * g[1] + b[1], (v[i][2] - mean) * rstd * g[2] + b[2], (v[i][3] - mean) * rstd * g[3] + b[3]);
    }
  }
}

// activation on an fp32 GEMM output -> operand pair. MODE 0: exact-erf GELU, 1: quick-GELU (in [rows][N] -> out [rows][N]);
// MODE 2: SwiGLU on the gate/up projection whose COLUMNS are interleaved in blocks of 16 (engine.interleave_gate_up: gate[16b..16b+15]
// at columns 32b..32b+15, up[...] at 32b+16..32b+31): in [rows][2 * Nout] -> out [rows][Nout], out[c] = silu(gate[c]) * up[c]
template <int MODE>
__global__ void act_pair_kernel(const float* __restrict__ in, int ldin, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int rows, int Nout) {
  const int per_row = Nout >> 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * per_row) return;
  const size_t r = i / per_row;
  const int c = (int)(i % per_row) * 4;
  float a[4];
  if (MODE == 2) {
    const int col = (c >> 4) * 32 + (c & 15);
    const f32x4 g = *(const f32x4*)(in + r * ldin + col);
    const f32x4 u = *(const f32x4*)(in + r * ldin + col + 16);
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = vt_silu(g[k]) * u[k];
  } else {
    const f32x4 v = *(const f32x4*)(in + r * ldin + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = (MODE == 0) ? vt_gelu_erf(v[k]) : vt_quick_gelu(v[k]);
  }
  store_pair4(hi + r * Nout + c, lo + r * Nout + c, a[0], a[1], a[2], a[3]);
}

// fp32 [rows][D] -> operand pair; with G2 > 0 the CLS row of every (G2 + 1)-row frame is dropped (feature_select of the towers)
__global__ void f32_to_pair_kernel(const float* __restrict__ x, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, size_t out_rows, int G2, int D) {
  const int per_row = D >> 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= out_rows * per_row) return;
  const size_t r = i / per_row;
  const int c = (int)(i % per_row) * 4;
  const size_t src = G2 > 0 ? (r / G2) * (G2 + 1) + 1 + (r % G2) : r;
  const f32x4 v = *(const f32x4*)(x + src * D + c);
  store_pair4(hi + r * D + c, lo + r * D + c, v[0], v[1], v[2], v[3]);
}

// dst fp32 += f32(a) for an operand-format a (the low half of an embedding pair joining the residual stream)
__global__ void add_op_to_f32_kernel(float* __restrict__ dst, const bf16_t* __restrict__ a, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const u32x2 w = *(const u32x2*)(a + i * 4);
  f32x4 v = *(const f32x4*)(dst + i * 4);
  v[0] += oplo_to_f32(w.x);
  v[1] += ophi_to_f32(w.x);
  v[2] += oplo_to_f32(w.y);
  v[3] += ophi_to_f32(w.y);
  *(f32x4*)(dst + i * 4) = v;
}

// many rows (prefill): persistent waves, each walking rows w, w + stride, ...; the NEXT row's 16-byte loads are issued before the
// current row is reduced and stored, so on the whole chip the read stream of one row overlaps the write stream of the previous
// one (the one-row-per-wave kernel above runs as two chip-wide phases: every wave reads, then every wave writes).
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_stream_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             bf16_t* __restrict__ y, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 g[NCH], cur[NCH], nxt[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + i * 64) * 4;
    g[i] = (c < D) ? *(const f32x4*)(w + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    cur[i] = (c < D) ? *(const f32x4*)(x + (size_t)row * D + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float inv_d = 1.0f / (float)D;
  for (; row < rows; row += stride) {
    const int nrow = row + stride;
    if (nrow < rows) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (lane + i * 64) * 4;
        nxt[i] = (c < D) ? *(const f32x4*)(x + (size_t)nrow * D + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the next row's loads ahead of this row's arithmetic and stores
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) sq += cur[i][0] * cur[i][0] + cur[i][1] * cur[i][1] + cur[i][2] * cur[i][2] + cur[i][3] * cur[i][3];
    const float rstd = rsqrtf(wave_sum(sq) * inv_d + eps);
    bf16_t* yr = y + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (lane + i * 64) * 4;
      if (c < D) {
        u32x2 o;
        o.x = pack_op2(cur[i][0] * rstd * g[i][0], cur[i][1] * rstd * g[i][1]);
        o.y = pack_op2(cur[i][2] * rstd * g[i][2], cur[i][3] * rstd * g[i][3]);
        *(u32x2*)(yr + c) = o;
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) cur[i] = nxt[i];
  }
}

// folded RMSNorm, tile-GEMM flavour: the residual GEMMs leave per-row partial sums of x^2, one per 32-column group, group-major
// ([np][ldp]); this turns them into the per-row factor the consumer GEMM's epilogue multiplies its accumulators with. One
// thread per (row, 1/16 of the groups): 16 rows per block (64-byte row segments per load), every thread's loads independent.
__global__ __launch_bounds__(256) void rowscale_finalize_kernel(const float* __restrict__ partials, int np, int ldp, int rows,
                                                                float inv_dim, float eps, float* __restrict__ out) {
  __shared__ float red[16][17];
  const int r = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int row = blockIdx.x * 16 + r;
  float s = 0.f;
  if (row < rows) {
#pragma unroll 8
    for (int g = q; g < np; g += 16) s += partials[(size_t)g * ldp + row];
  }
  red[q][r] = s;
  __syncthreads();
  if (q == 0 && row < rows) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][r];
    out[row] = rsqrtf(t * inv_dim + eps);
  }
}

// few rows (decode steps): one 256-thread block per row so the row's 16 KiB are fetched by 4 waves at once
__global__ __launch_bounds__(256) void rmsnorm_row_block_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                                const float* __restrict__ w, bf16_t* __restrict__ y,
                                                                int D, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const float* xr = x + (size_t)(idx ? idx[row] : row) * D;
  f32x4 v[4];
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (threadIdx.x + i * 256) * 4;
    if (c < D) {
      v[i] = *(const f32x4*)(xr + c);
      sq += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D + eps);
  bf16_t* yr = y + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (threadIdx.x + i * 256) * 4;
    if (c < D) {
      const f32x4 g = *(const f32x4*)(w + c);
      u32x2 o;
      o.x = pack_op2(v[i][0] * rstd * g[0], v[i][1] * rstd * g[1]);
      o.y = pack_op2(v[i][2] * rstd * g[2], v[i][3] * rstd * g[3]);
      *(u32x2*)(yr + c) = o;
    }
  }
}

__global__ void gather_f32_to_bf16_kernel(const float* __restrict__ in, const int* __restrict__ idx,
                                          bf16_t* __restrict__ out, int rows, int D) {
  const int per_row = D >> 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * per_row) return;
  const int r = (int)(i / per_row), c = (int)(i % per_row) * 4;
  const int src = idx ? idx[r] : r;
  const f32x4 v = *(const f32x4*)(in + (size_t)src * D + c);
  u32x2 o;
  o.x = pack_op2(v[0], v[1]);
  o.y = pack_op2(v[2], v[3]);
  *(u32x2*)(out + (size_t)r * D + c) = o;
}

__global__ void drop_cls_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int F, int G2, int D) {
  const int per_row = D >> 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)F * G2 * per_row) return;
  const size_t r = i / per_row;
  const int c = (int)(i % per_row) * 4;
  const size_t f = r / G2, p = r % G2;
  const f32x4 v = *(const f32x4*)(x + (f * (G2 + 1) + 1 + p) * D + c);
  u32x2 o;
  o.x = pack_op2(v[0], v[1]);
  o.y = pack_op2(v[2], v[3]);
  *(u32x2*)(out + r * D + c) = o;
}

__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const u32x2 w = *(const u32x2*)(in + i * 4);
  f32x4 v = {oplo_to_f32(w.x), ophi_to_f32(w.x), oplo_to_f32(w.y), ophi_to_f32(w.y)};
  *(f32x4*)(out + i * 4) = v;
}

__global__ void add_f32_kernel(float* __restrict__ dst, const float* __restrict__ a, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += a[i];
}

}  // namespace

#define VT_NORM_DISPATCH(KERN, D, ...)                                                                   \
  do {                                                                                                   \
    const int nch = cdiv(D, 256);                                                                        \
    if (nch <= 1) hipLaunchKernelGGL((KERN<1>), grid, dim3(256), 0, s, __VA_ARGS__);                      \
    else if (nch <= 2) hipLaunchKernelGGL((KERN<2>), grid, dim3(256), 0, s, __VA_ARGS__);                 \
    else if (nch <= 4) hipLaunchKernelGGL((KERN<4>), grid, dim3(256), 0, s, __VA_ARGS__);                 \
    else if (nch <= 8) hipLaunchKernelGGL((KERN<8>), grid, dim3(256), 0, s, __VA_ARGS__);                 \
    else hipLaunchKernelGGL((KERN<16>), grid, dim3(256), 0, s, __VA_ARGS__);                              \
  } while (0)

int vt_layernorm_launch(float* x, const float* temb, int T, int tokens_per_frame, const float* gamma,
                        const float* beta, bf16_t* y, int rows, int D, float eps, hipStream_t s) {
  VT_REQUIRE(x && gamma && beta && y, "vt_layernorm: null pointer");
  VT_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 4096, "vt_layernorm: D=%d must be a multiple of 4, <= 4096", D);
  if (temb) VT_REQUIRE(T > 0 && tokens_per_frame > 0, "vt_layernorm: temporal embedding needs T and tokens_per_frame");
  dim3 grid(cdiv(rows, 4));
  VT_NORM_DISPATCH(layernorm_kernel, D, x, temb, T, tokens_per_frame, gamma, beta, y, rows, D, eps);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_rmsnorm_launch(const float* x, const int* idx, const float* w, bf16_t* y, int rows, int D, float eps,
                      hipStream_t s) {
  VT_REQUIRE(x && w && y, "vt_rmsnorm: null pointer");
  VT_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 4096, "vt_rmsnorm: D=%d must be a multiple of 4, <= 4096", D);
  // up to ~2000 rows (decode steps, single-image prompts): one 256-thread block per row -- with one WAVE per row a 1088-row launch
  // keeps only 1088 x 16 loads in flight and took 13.6 us for 27 MB (2 TB/s)
  if (rows < 2048) {
    hipLaunchKernelGGL(rmsnorm_row_block_kernel, dim3(rows), dim3(256), 0, s, x, idx, w, y, D, eps);
    VT_LAUNCH_CHECK();
    return VT_OK;
  }
  if (!idx && rows >= 2048) {   // prefill-sized: persistent waves with the next row prefetched (two waves per SIMD, 256 CUs)
    dim3 grid(512);
    VT_NORM_DISPATCH(rmsnorm_stream_kernel, D, x, w, y, rows, D, eps);
    VT_LAUNCH_CHECK();
    return VT_OK;
  }
  dim3 grid(cdiv(rows, 4));
  VT_NORM_DISPATCH(rmsnorm_kernel, D, x, idx, w, y, rows, D, eps);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_rmsnorm_hilo_launch(const float* x, const int* idx, const float* w, bf16_t* y, bf16_t* ylo, int rows, int D, float eps, hipStream_t s) {
  VT_REQUIRE(x && w && y && ylo, "vt_rmsnorm_hilo: null pointer");
  VT_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 4096, "vt_rmsnorm_hilo: D=%d must be a multiple of 4, <= 4096", D);
  dim3 grid(cdiv(rows, 4));
  VT_NORM_DISPATCH(rmsnorm_hilo_kernel, D, x, idx, w, y, ylo, rows, D, eps);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_layernorm_hilo_launch(const float* x, const float* gamma, const float* beta, bf16_t* y, bf16_t* ylo, int rows, int D, float eps,
                             hipStream_t s) {
  VT_REQUIRE(x && gamma && beta && y && ylo, "vt_layernorm_hilo: null pointer");
  VT_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 4096, "vt_layernorm_hilo: D=%d must be a multiple of 4, <= 4096", D);
  dim3 grid(cdiv(rows, 4));
  VT_NORM_DISPATCH(layernorm_hilo_kernel, D, x, gamma, beta, y, ylo, rows, D, eps);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_act_pair_launch(const float* in, int ldin, bf16_t* hi, bf16_t* lo, int rows, int Nout, int mode, hipStream_t s) {
  VT_REQUIRE(in && hi && lo && rows > 0 && Nout > 0, "vt_act_pair: bad arguments");
  VT_REQUIRE(mode >= 0 && mode <= 2 && (Nout % (mode == 2 ? 16 : 4)) == 0 && (ldin % 4) == 0, "vt_act_pair: mode %d, Nout %d, ldin %d", mode, Nout, ldin);
  const size_t n = (size_t)rows * (Nout / 4);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (mode == 0) hipLaunchKernelGGL(act_pair_kernel<0>, grid, block, 0, s, in, ldin, hi, lo, rows, Nout);
  else if (mode == 1) hipLaunchKernelGGL(act_pair_kernel<1>, grid, block, 0, s, in, ldin, hi, lo, rows, Nout);
  else hipLaunchKernelGGL(act_pair_kernel<2>, grid, block, 0, s, in, ldin, hi, lo, rows, Nout);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_f32_to_pair_launch(const float* x, bf16_t* hi, bf16_t* lo, size_t out_rows, int G2, int D, hipStream_t s) {
  VT_REQUIRE(x && hi && lo && out_rows > 0 && D % 4 == 0 && G2 >= 0, "vt_f32_to_pair: bad arguments");
  const size_t n = out_rows * (D / 4);
  hipLaunchKernelGGL(f32_to_pair_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, hi, lo, out_rows, G2, D);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_add_op_to_f32_launch(float* dst, const bf16_t* a, size_t n, hipStream_t s) {
  VT_REQUIRE(dst && a && n % 4 == 0, "vt_add_op_to_f32: n must be a multiple of 4");
  const size_t n4 = n / 4;
  if (!n4) return VT_OK;
  hipLaunchKernelGGL(add_op_to_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, dst, a, n4);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_rowscale_finalize_launch(const float* partials, int np, int ldp, int rows, float inv_dim, float eps, float* out,
                                hipStream_t s) {
  VT_REQUIRE(partials && out && rows > 0 && np > 0 && ldp >= rows, "vt_rowscale_finalize: bad arguments (rows=%d np=%d ldp=%d)", rows, np, ldp);
  hipLaunchKernelGGL(rowscale_finalize_kernel, dim3(cdiv(rows, 16)), dim3(256), 0, s, partials, np, ldp, rows, inv_dim, eps, out);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_gather_f32_to_bf16_launch(const float* in, const int* idx, bf16_t* out, int rows, int D, hipStream_t s) {
  VT_REQUIRE(in && out && rows > 0 && D % 4 == 0, "vt_gather_f32_to_bf16: bad arguments");
  const size_t n = (size_t)rows * (D / 4);
  hipLaunchKernelGGL(gather_f32_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, idx, out, rows, D);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_bf16_to_f32_launch(const bf16_t* in, float* out, size_t n, hipStream_t s) {
  VT_REQUIRE(in && out && n % 4 == 0, "vt_bf16_to_f32: n must be a multiple of 4");
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, in, out, n4);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_add_f32_launch(float* dst, const float* a, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(add_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, a, n);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_drop_cls_launch(const float* x, bf16_t* out, int F, int G2, int D, hipStream_t s) {
  VT_REQUIRE(x && out && D % 4 == 0, "vt_drop_cls: bad arguments");
  const size_t n = (size_t)F * G2 * (D / 4);
  hipLaunchKernelGGL(drop_cls_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out, F, G2, D);
  VT_LAUNCH_CHECK();
  return VT_OK;
}
