// vt_gemm.hip -- bf16 x bf16 -> fp32-accumulate GEMM family for gfx950 (MI355X).
//
//   C[M,N] (+)= epilogue( A[M,K] . W[N,K]^T + bias[N] )
//
// A is an activation matrix (row-major, K contiguous), W a torch.nn.Linear weight ([out,in],
// K contiguous) -- so both operands stream K-contiguous and no transpose is ever materialised.
// Replaces every Linear on the hot path (SURVEY.md 8(a) rows V2,V4,V5,V6,P1,L3,L4,L6):
//   CLIPAttention q/k/v/out_proj, CLIPMLP fc1/fc2 (transformers 4.31, used at
//   reference vitron/model/multimodal_encoder/languagebind/video/modeling_video.py:69,71,81),
//   mm_projector (reference vitron/model/multimodal_projector/builder.py:33-51),
//   LlamaAttention / LlamaMLP / lm_head (reference vitron/model/language_model/llava_llama.py:49,91-102).
//
// Kernel structure ("tile" kernel):
//   * BM x BN output tile per workgroup, BK = 64, WAVES_M x WAVES_N wavefronts (64 lanes each),
//     every wave owns a (BM/WAVES_M) x (BN/WAVES_N) sub-tile as MI x NI fragments of
//     v_mfma_f32_16x16x32_bf16.
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B per lane, 1 KiB per wave-instruction),
//     two LDS stages, one barrier per K tile; the next tile's DMA is issued before the current
//     tile's MFMAs.
//   * LDS image is lane-linear (a DMA requirement); bank conflicts of the ds_read_b128 fragment
//     reads are removed by XOR-swizzling the 16-B chunk index with ((row>>1)&7) on the *source*
//     address and again on the read address (same involution both sides).
//   * MFMA operands are passed swapped (W fragment as "A", activation fragment as "B") so each
//     lane ends up with 4 consecutive output columns of one output row -> 8/16-byte stores and
//     lane-local epilogues (bias, GELU, residual, SwiGLU pairing).
//   * 1-D grid, XCD-aware + grouped tile order so the 32 CUs of an XCD work on neighbouring tiles.
//
// "skinny" kernels: M <= 16 rows (decode steps, region MLP, last-position lm_head) and 17..32 rows (larger decode batches,
// short prompts): pure weight streaming through a per-wave LDS-DMA ring into 16x16x32 MFMAs, split-K inside the block,
// optional folded RMSNorm. HBM-bound. The dispatcher (vt_gemm_launch / vt_gemm_pick_cfg / vt_gemm_resid_launch) at the end of
// the file picks between them, the 256x256 ping-pong kernel of vt_gemm8.hip (whole rounds of tiles), the M-split, the two-pass
// split-K and the small tiles -- every rule there carries the measurement it came from.
#include <algorithm>

#include "vt_common.h"
#include "vt_kernels.h"

namespace {

struct GemmP {
  const bf16_t* A;
  const bf16_t* W;
  void* C;
  const float* bias;
  int M, N, K;
  int lda, ldw, ldc;
  VtGemmNormFuse nf;   // folded RMSNorm / row scale (vt_kernels.h): skinny LDS-DMA kernel (decode flavour), tile kernel (row_scale, out_*)
};

__device__ __forceinline__ float gelu_erf(float x) { return vt_gelu_erf(x); }
__device__ __forceinline__ float quick_gelu(float x) { return vt_quick_gelu(x); }
__device__ __forceinline__ float silu(float x) { return vt_silu(x); }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_bt_kernel(GemmP p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BK = 64;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_PER_WAVE = (BM / 8) / NW, B_PER_WAVE = (BN / 8) / NW;
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split evenly over waves");
  static_assert(EPI != VT_EPI_SWIGLU_BF16 || (NI % 2 == 0), "SwiGLU pairs 16-column fragments");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- tile order: XCD remap, then groups of 8 M-tiles swept along N -------------------------
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  const int sid = xcd_remap(blockIdx.x, nwg);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int gid = sid / per_group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (sid % per_group) % gsz;
  const int tn = (sid % per_group) / gsz;
  const int bm0 = tm * BM, bn0 = tn * BN;

  // ---- per-lane DMA source pointers (k = 0), advanced by BK elements per K tile -----------------
  const int lrow = lane >> 3;   // row inside an 8-row DMA piece
  const int lchk = lane & 7;    // 16-B chunk position inside the 128-B LDS row
  const bf16_t* a_src[A_PER_WAVE];
  const bf16_t* b_src[B_PER_WAVE];
#pragma unroll
  for (int i = 0; i < A_PER_WAVE; ++i) {
    const int row = (wave * A_PER_WAVE + i) * 8 + lrow;
    const int csrc = lchk ^ ((row >> 1) & 7);
    const int grow = min(bm0 + row, p.M - 1);
    a_src[i] = p.A + (size_t)grow * p.lda + csrc * 8;
  }
#pragma unroll
  for (int i = 0; i < B_PER_WAVE; ++i) {
    const int row = (wave * B_PER_WAVE + i) * 8 + lrow;
    const int csrc = lchk ^ ((row >> 1) & 7);
    const int grow = min(bn0 + row, p.N - 1);
    b_src[i] = p.W + (size_t)grow * p.ldw + csrc * 8;
  }

  auto stage = [&](int buf) {
    char* base = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
      glds16(a_src[i], base + (wave * A_PER_WAVE + i) * 1024);
      a_src[i] += BK;
    }
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
      glds16(b_src[i], base + A_BYTES + (wave * B_PER_WAVE + i) * 1024);
      b_src[i] += BK;
    }
  };

  // ---- fragment read offsets -----------------------------------------------------------------
  // lane reads row (lane&15), logical chunk kk*4 + (lane>>4); swizzle f(row) = (row>>1)&7 = (lane>>1)&7
  const int f = (lane >> 1) & 7;
  const int frag_off0 = (lane & 15) * 128 + ((((lane >> 4)) ^ f) << 4);       // kk = 0
  const int frag_off1 = (lane & 15) * 128 + ((((lane >> 4) | 4) ^ f) << 4);   // kk = 1
  const int a_frag_base = wm * WTM * 128;
  const int b_frag_base = A_BYTES + wn * WTN * 128;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = p.K / BK;
  stage(0);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) stage(cur ^ 1);
    const char* sb = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = kk ? frag_off1 : frag_off0;
      bf16x8 af[MI], bfr[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *(const bf16x8*)(sb + a_frag_base + mi * 2048 + fo);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bfr[ni] = *(const bf16x8*)(sb + b_frag_base + ni * 2048 + fo);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = VT_MFMA_16x16x32(bfr[ni], af[mi], acc[mi][ni]);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n0..n0+3], m = ..+(lane&15), n0 = ..+(lane>>4)*4 ----------------
  // folded RMSNorm (VtGemmNormFuse, rows > 64): consumer = scale the row's accumulators by row_scale[m] first; producer
  // (residual epilogue) = also store bf16(x_new * out_w[n]) and the row's sum of x_new^2 over each 32-column group
  const float* const row_scale = p.nf.row_scale;
  float* const out_partials = (EPI == VT_EPI_F32_RESID) ? p.nf.out_partials : nullptr;
  const int m_lane = bm0 + wm * WTM + (lane & 15);
  const int n_lane = bn0 + wn * WTN + ((lane >> 4) << 2);
  float rsv[MI];   // fetched before the first store: a load inside the row loop would queue behind the previous row's stores
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) rsv[mi] = row_scale ? row_scale[min(m_lane + mi * 16, p.M - 1)] : 1.f;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m_lane + mi * 16;
    if (m >= p.M) continue;
    const float rs = rsv[mi];
    if constexpr (EPI == VT_EPI_SWIGLU_BF16) {
      // W rows are interleaved in blocks of 16: [gate 16 | up 16 | gate 16 | up 16 ...]
      bf16_t* crow = (bf16_t*)p.C + (size_t)m * p.ldc;
#pragma unroll
      for (int ni = 0; ni < NI; ni += 2) {
        const int n = n_lane + ni * 16;  // column of the gate fragment in the interleaved space
        if (n >= p.N) continue;
        const int no = ((bn0 + wn * WTN + ni * 16) >> 1) + ((lane >> 4) << 2);
        const f32x4 g = acc[mi][ni] * rs, u = acc[mi][ni + 1] * rs;
        u32x2 o;
        o.x = pack_op2(silu(g[0]) * u[0], silu(g[1]) * u[1]);
        o.y = pack_op2(silu(g[2]) * u[2], silu(g[3]) * u[3]);
        *(u32x2*)(crow + no) = o;
      }
    } else {
      float ss = 0.f;   // producer: sum of x_new^2 over the current 32-column group (two fragments)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n_lane + ni * 16;
        if (n < p.N) {
          f32x4 v = acc[mi][ni] * rs;
          if (p.bias) {
            const f32x4 b = *(const f32x4*)(p.bias + n);
            v += b;
          }
          if constexpr (EPI == VT_EPI_BF16_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
          } else if constexpr (EPI == VT_EPI_BF16_QGELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
          } else if constexpr (EPI == VT_EPI_BF16_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          if constexpr (EPI == VT_EPI_F32_RESID) {
            float* c = (float*)p.C + (size_t)m * p.ldc + n;
            const f32x4 nv = *(const f32x4*)c + v;
            *(f32x4*)c = nv;
            if (out_partials) {
              const f32x4 w4 = *(const f32x4*)(p.nf.out_w + n);
              u32x2 o;
              o.x = pack_op2(nv[0] * w4[0], nv[1] * w4[1]);
              o.y = pack_op2(nv[2] * w4[2], nv[3] * w4[3]);
              *(u32x2*)(p.nf.out_xw + (size_t)m * p.nf.ld_xw + n) = o;
              ss += (nv[0] * nv[0] + nv[1] * nv[1]) + (nv[2] * nv[2] + nv[3] * nv[3]);
            }
          } else if constexpr (EPI == VT_EPI_F32) {
            *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = v;
          } else {
            u32x2 o;
            o.x = pack_op2(v[0], v[1]);
            o.y = pack_op2(v[2], v[3]);
            *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
          }
        }
        if constexpr (EPI == VT_EPI_F32_RESID) {
          if ((ni & 1) && out_partials) {   // the 4 lanes (lane>>4) of a row hold the group's 32 columns; N % 32 == 0
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const int grp = (bn0 + wn * WTN + (ni - 1) * 16) >> 5;
            if ((lane >> 4) == 0 && grp < p.nf.out_np) out_partials[(size_t)grp * p.nf.out_ldp + m] = ss;
            ss = 0.f;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// skinny GEMM, register-operand variant (VT_GEMM_CFG_SKINNY_REG; the fallback when K % 64 != 0): M <= 16 rows (decode
// steps, region MLP, last-position lm_head), pure weight streaming on the matrix cores.
//   * a block of 8 waves owns 16 output columns (32 = one gate block + one up block for SwiGLU) and splits K eight
//     ways; partial 16x16 tiles are reduced through 8 KiB of LDS at the end,
//   * the weight rows are the MFMA "A" operand straight from HBM: lane (i = lane&15, g = lane>>4) loads the 16 bytes
//     W[n0+i][k + 8g .. +8) -- 64 contiguous bytes per row and instruction, UNR K-steps (32 wide) in flight per lane
//     and a second group prefetched before the first is consumed,
//   * the activations are the "B" operand, 16-B loads from L2 by the lanes whose row (lane&15) < M only,
//   * v_mfma_f32_16x16x32_bf16: lane ends with D[n = 4*(lane>>4)+r][m = lane&15].
// Roofline: HBM (2*N*K weight bytes per launch, read exactly once); the MFMA pipe idles (one 16x16x32 per KiB).
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512) void gemm_skinny_mfma_kernel(GemmP p) {
  constexpr int NT = (EPI == VT_EPI_SWIGLU_BF16) ? 2 : 1;   // 16-row weight tiles per block
  constexpr int NWAVE = 8;
  constexpr int UNR = (NT == 2) ? 4 : 8;
  __shared__ float red[NWAVE][NT][256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int n_base = blockIdx.x * 16 * NT;
  const int steps = (p.K + 31) >> 5;
  const int s0 = (int)((long)steps * wave / NWAVE), s1 = (int)((long)steps * (wave + 1) / NWAVE);
  const bf16_t* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wp[t] = p.W + (size_t)min(n_base + t * 16 + i, p.N - 1) * p.ldw + g * 8;
  const bf16_t* xp = p.A + (size_t)min(i, p.M - 1) * p.lda + g * 8;
  const bool xact = i < p.M;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 wa[UNR][NT], xa[UNR], wb[UNR][NT], xb[UNR];
#define SK_LOAD(W_, X_, S_)                                                                               \
  _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                       \
    const int st = (S_) + u;                                                                              \
    const bool ok = st < s1 && (st * 32 + g * 8) < p.K;                                                   \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                        \
      W_[u][t] = ok ? __builtin_nontemporal_load((const u32x4*)(wp[t] + (size_t)st * 32)) : (u32x4){0u, 0u, 0u, 0u}; \
    X_[u] = (ok && xact) ? *(const u32x4*)(xp + (size_t)st * 32) : (u32x4){0u, 0u, 0u, 0u};              \
  }
#define SK_MFMA(W_, X_)                                                                                   \
  _Pragma("unroll") for (int u = 0; u < UNR; ++u)                                                         \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                        \
      acc[t] = VT_MFMA_16x16x32(__builtin_bit_cast(bf16x8, W_[u][t]), __builtin_bit_cast(bf16x8, X_[u]), acc[t]);
  SK_LOAD(wa, xa, s0);
  for (int s = s0; s < s1; s += 2 * UNR) {
    SK_LOAD(wb, xb, s + UNR);
    SK_MFMA(wa, xa);
    SK_LOAD(wa, xa, s + 2 * UNR);
    SK_MFMA(wb, xb);
  }
#undef SK_LOAD
#undef SK_MFMA

#pragma unroll
  for (int t = 0; t < NT; ++t) *(f32x4*)&red[wave][t][lane * 4] = acc[t];
  __syncthreads();
  if (threadIdx.x >= 256) return;
  const int m = threadIdx.x >> 4, nn = threadIdx.x & 15;
  if (m >= p.M) return;
  const int e = (((nn >> 2) * 16 + m) << 2) + (nn & 3);   // lane (m, g = nn>>2) register nn&3
  float v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    v[t] = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) v[t] += red[w][t][e];
  }
  if constexpr (EPI == VT_EPI_SWIGLU_BF16) {
    const int q = blockIdx.x * 16 + nn;                      // output column; gate row n_base+nn, up row n_base+16+nn
    if (n_base + 16 + nn < p.N) ((bf16_t*)p.C)[(size_t)m * p.ldc + q] = f32_to_op(silu(v[0]) * v[NT - 1]);
  } else {
    const int n = n_base + nn;
    if (n >= p.N) return;
    float r = v[0] + (p.bias ? p.bias[n] : 0.f);
    if constexpr (EPI == VT_EPI_BF16_GELU) r = gelu_erf(r);
    if constexpr (EPI == VT_EPI_BF16_QGELU) r = quick_gelu(r);
    if constexpr (EPI == VT_EPI_BF16_RELU) r = fmaxf(r, 0.f);
    if constexpr (EPI == VT_EPI_F32_RESID) {
      float* c = (float*)p.C + (size_t)m * p.ldc + n;
      *c = *c + r;
    } else if constexpr (EPI == VT_EPI_F32) {
      ((float*)p.C)[(size_t)m * p.ldc + n] = r;
    } else {
      ((bf16_t*)p.C)[(size_t)m * p.ldc + n] = f32_to_op(r);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// skinny GEMM, LDS-DMA weight stream (default for M <= 16 when K % 64 == 0):
//   * a block of NWAVE waves owns 16*NT weight rows (NT = 2 for SwiGLU: one gate block + one up block) and splits K;
//     every wave runs ALONE on a private ring of R slots in LDS -- no block barrier until the final split-K reduce,
//   * one slot = one 64-wide K step: NT*2 `global_load_lds` instructions of 8 rows x 128 B for the weights (nt policy:
//     read once) + 1 (M <= 8) or 2 for the activation rows, all XOR-swizzled on the source side exactly like the tile
//     kernel, so the 16-B MFMA fragment reads are conflict-free,
//   * counted `s_waitcnt vmcnt` keeps (R-1) K steps of DMA in flight per wave while the oldest slot is consumed by
//     two v_mfma_f32_16x16x32_bf16 per weight tile (weights = "A" operand, activations = "B": lane ends with
//     D[n = 4*(lane>>4)+r][m = lane&15]); activation rows >= M hold whatever the ring held -- they only reach output
//     columns m >= M, which are never stored.
// Roofline: HBM (2*N*K weight bytes per launch, read exactly once).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16_nt(const void* gsrc, void* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 2);
}
template <int N>
__device__ __forceinline__ void vmcnt_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI, int XI, int NWAVE, int R>
__global__ __launch_bounds__(64 * NWAVE) void gemm_skinny_dma_kernel(GemmP p) {
  constexpr int NT = (EPI == VT_EPI_SWIGLU_BF16) ? 2 : 1;   // 16-row weight tiles per block
  constexpr int PS = NT * 2 + XI;                            // DMA instructions per K step
  constexpr int SLOT = PS * 1024;
  static_assert((R & (R - 1)) == 0 && (R - 1) * PS <= 63, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[NWAVE][NT][256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = smem + wave * (R * SLOT);
  const int n_base = blockIdx.x * 16 * NT;
  const int steps = p.K >> 6;
  const int s0 = (int)((long)steps * wave / NWAVE), s1 = (int)((long)steps * (wave + 1) / NWAVE);
  const int n = s1 - s0;

  // DMA sources at K step 0: lane -> (row = lane>>3 of an 8-row piece, 16-B chunk lane&7, swizzled on the source side)
  const int lrow = lane >> 3, lchk = lane & 7;
  const bf16_t* wsrc[NT * 2];
#pragma unroll
  for (int q = 0; q < NT * 2; ++q) {
    const int row = (q & 1) * 8 + lrow;                      // row inside the 16-row tile
    const int csrc = lchk ^ ((row >> 1) & 7);
    wsrc[q] = p.W + (size_t)min(n_base + (q >> 1) * 16 + row, p.N - 1) * p.ldw + csrc * 8;
  }
  const bf16_t* xsrc[XI];
#pragma unroll
  for (int q = 0; q < XI; ++q) {
    const int row = q * 8 + lrow;
    const int csrc = lchk ^ ((row >> 1) & 7);
    xsrc[q] = p.A + (size_t)min(row, p.M - 1) * p.lda + csrc * 8;
  }
  auto issue = [&](int step, int slot) {
    char* base = ring + slot * SLOT;
    const size_t ko = (size_t)step * 64;
#pragma unroll
    for (int q = 0; q < NT * 2; ++q) glds16_nt(wsrc[q] + ko, base + q * 1024);
#pragma unroll
    for (int q = 0; q < XI; ++q) glds16(xsrc[q] + ko, base + (NT * 2 + q) * 1024);
  };
  const int f = (lane >> 1) & 7;
  const int frag_off0 = (lane & 15) * 128 + (((lane >> 4) ^ f) << 4);
  const int frag_off1 = (lane & 15) * 128 + ((((lane >> 4) | 4) ^ f) << 4);

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int r = 0; r < R; ++r)
    if (r < n) issue(s0 + r, r);
  // folded RMSNorm, consumer side: thread (row m = tid>>4, part tid&15) fetches its in_n/16 (<= 32) partial sums of squares
  // right behind the ring's first fills, so the look-up rides under the weight stream instead of sitting on the block's tail
  f32x4 rsv[8];
  float rs_keep[8];
  if (p.nf.in_partials) {   // branch-free: clamped addresses, out-of-range quads get weight 0 -> all 8 loads go out back to back
    const int cnt = p.nf.in_n >> 4;
    const int row = min((int)(threadIdx.x >> 4) & 15, p.M - 1);
    const float* pp = p.nf.in_partials + (size_t)row * p.nf.in_n + (threadIdx.x & 15) * cnt;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      rsv[q] = *(const f32x4*)(pp + min(4 * q, cnt - 4));
      rs_keep[q] = (4 * q < cnt) ? 1.f : 0.f;
    }
  }
  for (int i = 0; i < n; ++i) {
    if (i + R <= n) vmcnt_wait<(R - 1) * PS>(); else vmcnt_wait<0>();
    const char* sb = ring + (i & (R - 1)) * SLOT;
    bf16x8 wf[NT][2], xf[2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      wf[t][0] = *(const bf16x8*)(sb + t * 2048 + frag_off0);
      wf[t][1] = *(const bf16x8*)(sb + t * 2048 + frag_off1);
    }
    xf[0] = *(const bf16x8*)(sb + NT * 2048 + frag_off0);   // XI == 1: rows >= 8 read past the 8-row piece (next slot / reduce
    xf[1] = *(const bf16x8*)(sb + NT * 2048 + frag_off1);   // buffer): garbage confined to output columns m >= 8 > M
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (i + R < n) issue(s0 + i + R, i & (R - 1));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t] = VT_MFMA_16x16x32(wf[t][0], xf[0], acc[t]);
      acc[t] = VT_MFMA_16x16x32(wf[t][1], xf[1], acc[t]);
    }
  }

#pragma unroll
  for (int t = 0; t < NT; ++t) *(f32x4*)&red[wave][t][lane * 4] = acc[t];
  __syncthreads();
  if (threadIdx.x >= 256) return;
  const int m = threadIdx.x >> 4, nn = threadIdx.x & 15;
  if (m >= p.M) return;
  const int e = (((nn >> 2) * 16 + m) << 2) + (nn & 3);   // lane (m, g = nn>>2), register nn&3
  float v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    v[t] = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) v[t] += red[w][t][e];
  }
  if (p.nf.in_partials) {   // folded RMSNorm, consumer side: the 16 threads of row m add up its partial sums of squares
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) ss += rs_keep[q] * ((rsv[q][0] + rsv[q][1]) + (rsv[q][2] + rsv[q][3]));
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) ss += __shfl_xor(ss, off, 16);
    const float rstd = rsqrtf(ss * p.nf.inv_dim + p.nf.eps);
#pragma unroll
    for (int t = 0; t < NT; ++t) v[t] *= rstd;
  }
  if constexpr (EPI == VT_EPI_SWIGLU_BF16) {
    const int q = blockIdx.x * 16 + nn;                      // output column; gate row n_base+nn, up row n_base+16+nn
    if (n_base + 16 + nn < p.N) ((bf16_t*)p.C)[(size_t)m * p.ldc + q] = f32_to_op(silu(v[0]) * v[NT - 1]);
  } else {
    const int nc = n_base + nn;
    if constexpr (EPI == VT_EPI_F32_RESID) {
      if (p.nf.out_partials) {   // folded RMSNorm, producer side (N % 16 == 0 checked by the launcher: all 16 threads are live)
        float* c = (float*)p.C + (size_t)m * p.ldc + nc;
        const float x = *c + v[0];
        *c = x;
        p.nf.out_xw[(size_t)m * p.nf.ld_xw + nc] = f32_to_op(x * p.nf.out_w[nc]);
        float ss = x * x;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) ss += __shfl_xor(ss, off, 16);
        if (nn == 0) p.nf.out_partials[(size_t)m * gridDim.x + blockIdx.x] = ss;
        return;
      }
    }
    if (nc >= p.N) return;
    float r = v[0] + (p.bias ? p.bias[nc] : 0.f);
    if constexpr (EPI == VT_EPI_BF16_GELU) r = gelu_erf(r);
    if constexpr (EPI == VT_EPI_BF16_QGELU) r = quick_gelu(r);
    if constexpr (EPI == VT_EPI_BF16_RELU) r = fmaxf(r, 0.f);
    if constexpr (EPI == VT_EPI_F32_RESID) {
      float* c = (float*)p.C + (size_t)m * p.ldc + nc;
      *c = *c + r;
    } else if constexpr (EPI == VT_EPI_F32) {
      ((float*)p.C)[(size_t)m * p.ldc + nc] = r;
    } else {
      ((bf16_t*)p.C)[(size_t)m * p.ldc + nc] = f32_to_op(r);
    }
  }
}

template <int BM, int BN, int WM, int WN, int EPI>
int launch_tile(const GemmP& p, hipStream_t s) {
  constexpr int smem = 2 * (BM + BN) * 64 * 2;
  auto kern = gemm_bt_kernel<BM, BN, WM, WN, EPI>;
  VT_LDS_ATTR_ONCE(kern, smem);
  const int nwg = cdiv(p.M, BM) * cdiv(p.N, BN);
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * WM * WN), smem, s, p);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

template <int EPI>
int launch_cfg(const GemmP& p, int cfg, hipStream_t s) {
  switch (cfg) {
    case VT_GEMM_CFG_128x128: return launch_tile<128, 128, 2, 2, EPI>(p, s);
    case VT_GEMM_CFG_256x128: return launch_tile<256, 128, 4, 2, EPI>(p, s);
    case VT_GEMM_CFG_256x256: return launch_tile<256, 256, 2, 4, EPI>(p, s);
    case VT_GEMM_CFG_64x128: return launch_tile<64, 128, 1, 4, EPI>(p, s);
    default: vt_set_error("vt_gemm: unknown tile config %d", cfg); return VT_ERR_ARG;
  }
}

template <int EPI>
int launch_skinny_mfma(const GemmP& p, hipStream_t s) {
  const int rows_per_block = (EPI == VT_EPI_SWIGLU_BF16) ? 32 : 16;
  hipLaunchKernelGGL(gemm_skinny_mfma_kernel<EPI>, dim3(cdiv(p.N, rows_per_block)), dim3(512), 0, s, p);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

// ------------------------------------------------------------------------------------------------
// The same weight stream for 17..32 activation rows (decode batches / short prompts of up to 32 rows): two 16-row MFMA column
// groups per weight tile; the 32 activation rows of a K step (4 KiB) ride the ring next to the weights (2 KiB per tile), all
// by LDS-DMA so that ONE counted vmcnt orders everything (mixing register loads into the same queue was tried: the counter
// does not order LDS-DMA against ordinary loads -- sporadic stale operands). One workgroup per CU at this slot size.
// WIDE: two weight tiles per block also for the plain epilogues (N >= 8192): halves the activation re-reads from L2.
// ------------------------------------------------------------------------------------------------
template <int EPI, int NWAVE, bool WIDE>
__global__ __launch_bounds__(64 * NWAVE) void gemm_skinny_dma32_kernel(GemmP p) {
  constexpr int NT = (EPI == VT_EPI_SWIGLU_BF16 || WIDE) ? 2 : 1;
  constexpr int R = 4, MT = 2, XI = 4;
  constexpr int PS = NT * 2 + XI;                            // DMA instructions per K step
  constexpr int SLOT = PS * 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[NWAVE][NT][MT][256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = smem + wave * (R * SLOT);
  const int n_base = blockIdx.x * 16 * NT;
  const int steps = p.K >> 6;
  const int s0 = (int)((long)steps * wave / NWAVE), s1 = (int)((long)steps * (wave + 1) / NWAVE);
  const int n = s1 - s0;
  const int lrow = lane >> 3, lchk = lane & 7;
  const bf16_t* wsrc[NT * 2];
#pragma unroll
  for (int q = 0; q < NT * 2; ++q) {
    const int row = (q & 1) * 8 + lrow;
    const int csrc = lchk ^ ((row >> 1) & 7);
    wsrc[q] = p.W + (size_t)min(n_base + (q >> 1) * 16 + row, p.N - 1) * p.ldw + csrc * 8;
  }
  const bf16_t* xsrc[XI];
#pragma unroll
  for (int q = 0; q < XI; ++q) {
    const int row = q * 8 + lrow;                            // activation row 0..31 (clamped: rows >= M only feed unused columns)
    const int csrc = lchk ^ (((row & 15) >> 1) & 7);         // swizzle by the row inside its 16-row MFMA group
    xsrc[q] = p.A + (size_t)min(row, p.M - 1) * p.lda + csrc * 8;
  }
  auto issue = [&](int step, int slot) {
    char* base = ring + slot * SLOT;
    const size_t ko = (size_t)step * 64;
#pragma unroll
    for (int q = 0; q < NT * 2; ++q) glds16_nt(wsrc[q] + ko, base + q * 1024);
#pragma unroll
    for (int q = 0; q < XI; ++q) glds16(xsrc[q] + ko, base + (NT * 2 + q) * 1024);
  };
  const int f = (lane >> 1) & 7;
  const int frag_off0 = (lane & 15) * 128 + (((lane >> 4) ^ f) << 4);
  const int frag_off1 = (lane & 15) * 128 + ((((lane >> 4) | 4) ^ f) << 4);

  f32x4 acc[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < MT; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (r < n) issue(s0 + r, r);
  for (int i = 0; i < n; ++i) {
    if (i + R <= n) vmcnt_wait<(R - 1) * PS>(); else vmcnt_wait<0>();
    const char* sb = ring + (i & (R - 1)) * SLOT;
    bf16x8 wf[NT][2], xf[MT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      wf[t][0] = *(const bf16x8*)(sb + t * 2048 + frag_off0);
      wf[t][1] = *(const bf16x8*)(sb + t * 2048 + frag_off1);
    }
#pragma unroll
    for (int g = 0; g < MT; ++g) {
      xf[g][0] = *(const bf16x8*)(sb + (NT + g) * 2048 + frag_off0);
      xf[g][1] = *(const bf16x8*)(sb + (NT + g) * 2048 + frag_off1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (i + R < n) issue(s0 + i + R, i & (R - 1));
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < MT; ++g) {
        acc[t][g] = VT_MFMA_16x16x32(wf[t][0], xf[g][0], acc[t][g]);
        acc[t][g] = VT_MFMA_16x16x32(wf[t][1], xf[g][1], acc[t][g]);
      }
  }

#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < MT; ++g) *(f32x4*)&red[wave][t][g][lane * 4] = acc[t][g];
  __syncthreads();
  if (threadIdx.x >= 256) return;
  const int ml = threadIdx.x >> 4, nn = threadIdx.x & 15;
  const int e = (((nn >> 2) * 16 + ml) << 2) + (nn & 3);
#pragma unroll
  for (int g = 0; g < MT; ++g) {
    const int m = g * 16 + ml;
    if (m >= p.M) continue;
    float v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      v[t] = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) v[t] += red[w][t][g][e];
    }
    if constexpr (EPI == VT_EPI_SWIGLU_BF16) {
      const int q = blockIdx.x * 16 + nn;
      if (n_base + 16 + nn < p.N) ((bf16_t*)p.C)[(size_t)m * p.ldc + q] = f32_to_op(silu(v[0]) * v[NT - 1]);
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int nc = n_base + t * 16 + nn;
        if (nc >= p.N) continue;
        float r = v[t] + (p.bias ? p.bias[nc] : 0.f);
        if constexpr (EPI == VT_EPI_BF16_GELU) r = gelu_erf(r);
        if constexpr (EPI == VT_EPI_BF16_QGELU) r = quick_gelu(r);
        if constexpr (EPI == VT_EPI_BF16_RELU) r = fmaxf(r, 0.f);
        if constexpr (EPI == VT_EPI_F32_RESID) {
          float* c = (float*)p.C + (size_t)m * p.ldc + nc;
          *c = *c + r;
        } else if constexpr (EPI == VT_EPI_F32) {
          ((float*)p.C)[(size_t)m * p.ldc + nc] = r;
        } else {
          ((bf16_t*)p.C)[(size_t)m * p.ldc + nc] = f32_to_op(r);
        }
      }
    }
  }
}

template <int EPI, bool WIDE>
int launch_skinny_dma32_w(const GemmP& p, hipStream_t s) {
  constexpr int NT = (EPI == VT_EPI_SWIGLU_BF16 || WIDE) ? 2 : 1;
  constexpr int NWAVE = 4;
  constexpr int smem = NWAVE * 4 * (NT * 2 + 4) * 1024;
  static_assert(smem + NWAVE * NT * 2 * 1024 <= 160 * 1024, "ring does not fit the 160 KiB LDS");
  auto kern = gemm_skinny_dma32_kernel<EPI, NWAVE, WIDE>;
  VT_LDS_ATTR_ONCE(kern, smem);
  hipLaunchKernelGGL(kern, dim3(cdiv(p.N, 16 * NT)), dim3(64 * NWAVE), smem, s, p);
  VT_LAUNCH_CHECK();
  return VT_OK;
}
template <int EPI>
int launch_skinny_dma32(const GemmP& p, hipStream_t s) {
  if (EPI != VT_EPI_SWIGLU_BF16 && p.N >= 8192) return launch_skinny_dma32_w<EPI, true>(p, s);
  return launch_skinny_dma32_w<EPI, false>(p, s);
}

template <int EPI, int XI, int NWAVE, int R>
int launch_skinny_dma_cfg(const GemmP& p, hipStream_t s) {
  constexpr int NT = (EPI == VT_EPI_SWIGLU_BF16) ? 2 : 1;
  constexpr int smem = NWAVE * R * (NT * 2 + XI) * 1024 + ((XI == 1) ? 1024 : 0);  // + slack for the XI == 1 over-read of the last slot
  static_assert(smem + NWAVE * NT * 1024 <= 160 * 1024, "ring does not fit the 160 KiB LDS");
  auto kern = gemm_skinny_dma_kernel<EPI, XI, NWAVE, R>;
  VT_LDS_ATTR_ONCE(kern, smem);
  hipLaunchKernelGGL(kern, dim3(cdiv(p.N, 16 * NT)), dim3(64 * NWAVE), smem, s, p);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

template <int EPI>
int launch_skinny(const GemmP& p, hipStream_t s, bool reg_operands) {
  // measured on MI355X (tools/skinny_bench.py): 4 waves x 4-slot rings beat 8x4, 4x8 and 2x8 on every decode shape
  if (p.M > 16) return launch_skinny_dma32<EPI>(p, s);   // 17..32 rows (K % 64 == 0 checked by the caller)
  if (reg_operands || (p.K % 64) != 0) return launch_skinny_mfma<EPI>(p, s);
#ifdef VT_ABLATIONS   // VT_SKINNY_VARIANT=<waves * 100 + ring slots> tools/skinny_bench.py (test library): other (waves, ring) shapes, M <= 8
  {
    static const int variant = getenv("VT_SKINNY_VARIANT") ? atoi(getenv("VT_SKINNY_VARIANT")) : 0;
    constexpr int NTv = (EPI == VT_EPI_SWIGLU_BF16) ? 2 : 1;
    if (variant && p.M <= 8) {
#define VT_SKV(NW, RR)                                                                               \
  if (variant == NW * 100 + RR) {                                                                    \
    if constexpr (NW * RR * (NTv * 2 + 1) * 1024 + 1024 + NW * NTv * 1024 <= 160 * 1024)             \
      return launch_skinny_dma_cfg<EPI, 1, NW, RR>(p, s);                                            \
  }
      VT_SKV(8, 4) VT_SKV(4, 8) VT_SKV(2, 4) VT_SKV(2, 8) VT_SKV(8, 8) VT_SKV(3, 4) VT_SKV(6, 4) VT_SKV(4, 16) VT_SKV(2, 16)
#undef VT_SKV
    }
  }
#endif
  if (p.M <= 8) return launch_skinny_dma_cfg<EPI, 1, 4, 4>(p, s);
  return launch_skinny_dma_cfg<EPI, 2, 4, 4>(p, s);
}

}  // namespace

int vt_gemm_pick_cfg(int M, int N, int K) {
  // Measured on MI355X (tools/gemm_bench.py, profiles/): the 256x256 ping-pong kernel (4-phase variant: 8-11 % faster
  // than the 8-phase one on every decoder shape) wins whenever its grid fills
  // whole rounds of the 256 CUs (one 128-KiB-LDS workgroup per CU); when the last round would be mostly empty the
  // 128x128 kernel (two workgroups per CU, 4x more tiles) quantises better. The register-pipelined variant
  // (VT_GEMM_CFG_256x256_RP) is 3-8% faster than the 8-phase kernel on isolated back-to-back launches but measured
  // equal (bf16 epilogue) to 9% slower (fp32 residual epilogue) inside the full prefill step, so it stays opt-in.
  if (M <= 64) return VT_GEMM_CFG_64x128;
  const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
  const long rounds = (t256 + 255) / 256;
  const double waste = 1.0 - (double)t256 / (double)(rounds * 256);
  if (vt_gemm_p8_supported(M, N, K) && waste <= 0.20) return VT_GEMM_CFG_256x256_P4;   // 4608x3072x1024 (216 tiles): 847 vs 706 TFLOP/s
  // 128x128 (two 64-KiB workgroups per CU) only pays with >= 2 full rounds of tiles and a long K loop; everything smaller
  // -- the 1024-row remainders of the M-split, the ViT / projector shapes with K = 1024 or N = 1024 -- measured 10-20 %
  // faster on 64x128 tiles (tools/gemm_bench.py: 1024x4096x4096 752 vs 651 TFLOP/s, 4608x1024x4096 681 vs 623)
  const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
  if (t128 >= 1024 && K >= 2048) return VT_GEMM_CFG_128x128;
  // 64x128 tiles run two workgroups per CU = 512 slots: when the grid spills a little over a whole number of rounds (M = 1088,
  // N = 4096: 544 tiles) the half-empty last round costs a full one; with a long K loop the 128x128 grid (288 tiles, all
  // resident) wins: 1088x4096x11008 723 vs 592 TFLOP/s (at K = 4096 the small tiles still win, 675 vs 629)
  const long t64 = (long)cdiv(M, 64) * cdiv(N, 128);
  const long r64 = (t64 + 511) / 512;
  if (K >= 8192 && t128 >= 256 && t128 <= 512 && (double)t64 / (double)(r64 * 512) < 0.7) return VT_GEMM_CFG_128x128;
  return VT_GEMM_CFG_64x128;
}

// ---- the tile-GEMM planner ---------------------------------------------------------------------------------------------------
// A big-tile kernel keeps ONE workgroup per CU, so its time is (rounds of 256 workgroups) x (tile height); the small tiles fill the
// chip at any size but run at about half the rate. The planner prices the candidates in units of "one round of 256-row tiles":
//   256-row tiles   ceil(tiles / 256)                       four-wave kernel for plain-store epilogues (sustained interleaved A/B,
//                                                            tools/gemm_ab.cpp, round 2: qkv 1408 vs 1331 TFLOP/s on the 4-phase
//                                                            ping-pong kernel, gate/up SwiGLU 1415 vs 1363, down_proj 1412 vs 1380,
//                                                            ViT qkv 876 vs 840); the ping-pong kernel where the epilogue is VALU-heavy
//                                                            (round 2: erf-GELU 366 vs 488 on the four-wave kernel's single wave per
//                                                            SIMD; with round 3's LDS-staged store epilogue the two are within 3 %:
//                                                            4096 x 4096 x 1024 erf-GELU 805 vs 819, quick-GELU 785 vs 813 TFLOP/s,
//                                                            profiles/r3_gemm_gelu_ab.jsonl) or carries a folded RMSNorm
//   320-row tiles   ceil(tiles320 / 256) x 1.25 x 1.04      four-wave kernel, 160x128 per wave: 5120 x 4096 is ONE round of 256 tiles
//                                                            instead of 1.25 (o_proj 1245 vs 1075 TFLOP/s for whole rounds + small-tile
//                                                            remainder, down_proj 1341 vs 1173, projector 4608x4096x4096 1329 vs 1131);
//                                                            on multi-round grids the tile itself is ~4 % slower (qkv 1420 vs 1405)
//   small tiles     ceil(tiles64x128 / 512) x 0.4545        two workgroups per CU at ~0.55 of the big tiles' rate (remainders of 512 /
//                                                            1024 rows x 4096: 0.40 / 0.39 of a big round measured; 2560 x 4096 x 4096: 827
//                                                            TFLOP/s on small tiles vs 1005 on 160 big tiles)
//   224-row tiles   ceil(tiles224 / 256) x 0.875 x 1.02     four-wave kernel, 112x128 per wave (round 3): 1088 rows are 5 tile rows of 224 (1120) instead
//                                                            of 5 of 256 (1280)
//   160x128 tiles   ceil(tiles160 / 256) x 0.42             four-wave kernel on a four-deep LDS ring (K % 256 == 0): one workgroup per CU, a
//                                                            round of them costs 0.41 of a big round (4616 x 1024 x 4096: 41 us, 935 TFLOP/s
//                                                            against 692 on 64x128 tiles; 1088 x 4096 x 4096: 906 vs 637)
//   M-split         whole rounds of 256-row tiles + the plan of the remaining rows (only with a long K loop)
// and takes the cheapest. VtGemmPlan.M1 > 0: rows [0, M1) go to cfg, the rest is planned again by the caller.
struct VtGemmPlan {
  int cfg;
  int M1;
  int N1 = 0;   // > 0: COLUMNS [0, N1) go to cfg (whole rounds of big tiles), the remaining columns are planned again (round 5)
};

static bool vt_epi_plain(int epi) { return epi == VT_EPI_BF16 || epi == VT_EPI_F32_RESID || epi == VT_EPI_F32 || epi == VT_EPI_SWIGLU_BF16; }

static int vt_gemm_256_variant(int epi, const VtGemmNormFuse* nf) {
  return (vt_epi_plain(epi) && !nf) ? VT_GEMM_CFG_256x256_W4 : VT_GEMM_CFG_256x256_P4;
}

static double vt_gemm_plan_cost(int M, int N, int K, int epi, const VtGemmNormFuse* nf, VtGemmPlan* plan, bool allow_split) {
  // 64x128 tiles: two workgroups per CU = 512 slots; a round of them moves 512 x 64 x 128 outputs at ~0.55 of the big tiles' rate
  const double small = (double)(((long)cdiv(M, 64) * cdiv(N, 128) + 511) / 512) * (512.0 * 64 * 128 / 65536.0 / 256.0 / 0.55);
  double best = small;
  *plan = VtGemmPlan{vt_gemm_pick_cfg(M, N, K), 0};
  if (plan->cfg == VT_GEMM_CFG_256x256_P4) plan->cfg = VT_GEMM_CFG_64x128;   // (the family is priced below)
  if (M <= 64 || !vt_gemm_p8_supported(M, N, K)) return best;
  const int tiles_n = cdiv(N, 256);
  const double c256 = (double)((cdiv(M, 256) * (long)tiles_n + 255) / 256);
  if (c256 < best) {
    best = c256;
    *plan = VtGemmPlan{vt_gemm_256_variant(epi, nf), 0};
  }
  if (!nf) {
    const double c320 = (double)((cdiv(M, 320) * (long)tiles_n + 255) / 256) * 1.25 * 1.04;
    if (c320 < best) {
      best = c320;
      *plan = VtGemmPlan{VT_GEMM_CFG_320x256_W4, 0};
    }
    // 224-row tiles (112x128 per wave): a round of them costs 7/8 of a 256-row round; wins where 256-row tiles pad the last tile row
    // heavily and the tile COUNT stays inside the same number of rounds (1088 rows: 5 tile rows either way; 4616 x 3072: 252 vs 228 tiles)
    const long t224 = cdiv(M, 224) * (long)tiles_n;
    const double c224 = (double)((t224 + 255) / 256) * 0.875 * 1.02;
    // Only where it PADS FEWER ROWS than the 256-row tile (1088 -> 1120 vs 1280; 4616 -> 4704 vs 4864): the 0.875 x 1.02 price was
    // measured on exactly those shapes (profiles/r3_gemm_tile224_ab.jsonl). A multiple of 256 (2048 rows: 160 tiles instead of 128,
    // 9 % padding) would otherwise switch kernels on an extrapolated model -- unmeasured shapes stay on the 256-row tile.
    // Round 5 measured the multiples of 256 the guard above used to exclude (profiles/r5_gemm_cfg_224px.jsonl, the reference-native 224 px
    // shapes): 2560 x 4096 x 11008 207 us on 192 tiles of 224 rows vs 233 on the ring kernel's 512 and 244 on 160 of 256 rows; 2560 x 4096 x 4096
    // 87 vs 92 vs 99; 768 x 12288 x 4096 80 vs 88 vs 94; 2048 x 4096 x 4096 72 vs 80 vs 85 -- the round model holds there too (a round of
    // 224-row tiles IS 7/8 as long whatever the padding), so the candidate is priced for every M.
    if (t224 >= 128 && c224 < best) {   // (under half a round the small tiles' finer grid is as good: not measured, left alone)
      best = c224;
      *plan = VtGemmPlan{VT_GEMM_CFG_224x256_W4, 0};
    }
    // (a grid under half a round with a short K loop stays on small tiles: 577 x 3072 x 1024, 96 tiles, 281 vs 339 TFLOP/s)
    const long t160 = cdiv(M, 160) * (long)cdiv(N, 128);
    if ((K % 256) == 0 && (t160 >= 128 || K >= 4096)) {
      // (round 5: every round AFTER the first costs 0.49, not 0.42 -- 2560 x 4096 x 4096, 512 tiles = two rounds: 92 us against 95 per round
      // of 256-row tiles; one round, 768 x 4096 x 4096: 43 us. The single-round price, and with it every choice of rounds 2-4, is unchanged)
      const long r160 = (t160 + 255) / 256;
      const double c160 = (double)r160 * 0.42 + (double)(r160 - 1) * 0.07;
      if (c160 < best) {
        best = c160;
        *plan = VtGemmPlan{VT_GEMM_CFG_160x128_W4, 0};
      }
    }
    // ONE round of 128x128 tiles (two 64-KiB workgroups per CU = 512 slots): where the grid fits that single round and covers at least half
    // of it, it beats a part-filled round of big tiles and two rounds of the ring kernel (round 5, profiles/r5_gemm_cfg_224px.jsonl:
    // 2048 x 4096 x 4096 67 us vs 72 on 224-row tiles / 80 on the ring; 2056 x 3072 x 1024 20.8 vs 22.9 / 25.2; 2048 x 4096 x 1024 + GELU
    // 23.4 vs 29.6 / 29.3; two rounds lose everywhere: 2560 x 4096 x 4096 125 vs 87). Priced at 0.70 of a big round (0.80 on a short K loop).
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
    if (t128 >= 256 && t128 <= 512) {
      const double c128 = K >= 2048 ? 0.70 : 0.80;
      if (c128 < best) {
        best = c128;
        *plan = VtGemmPlan{VT_GEMM_CFG_128x128, 0};
      }
    }
  }
  if (allow_split && K >= 2048 && M > 256) {
    int g = tiles_n, b = 256;
    while (b) {   // gcd(tiles_n, 256)
      const int t = g % b;
      g = b;
      b = t;
    }
    const long unit = 256L * (256 / g);   // rows per whole round
    const long M1 = (M / unit) * unit;
    if (M1 >= unit && M1 < M) {
      VtGemmPlan rest;
      const double c = (double)(M1 / 256 * tiles_n / 256) + vt_gemm_plan_cost(M - (int)M1, N, K, epi, nf, &rest, false) + 0.02;   // + a launch
      if (c < best) {
        best = c;
        *plan = VtGemmPlan{vt_gemm_256_variant(epi, nf), (int)M1};
      }
    }
  }
  // Column split (round 5): a few row blocks x many column tiles that spill just over whole rounds (768 x 22016: 3 x 86 = 258 tiles of 256
  // rows, two rounds for 1.008 rounds of work) -- the columns that fill WHOLE rounds go to the 256-row tiles, the remaining columns are planned
  // again (measured: 21504 columns 107 us + 512 columns on the ring 30 us = 137 against 158 for the best single grid).
  if (allow_split && !nf && K >= 2048 && vt_epi_plain(epi)) {
    const long rb = cdiv(M, 256);
    const long whole = ((long)rb * tiles_n) / 256;          // whole rounds the grid contains
    if (whole >= 1 && rb <= 256) {
      const long n1_tiles = whole * 256 / rb;                // column tiles that fit those rounds
      const long N1 = n1_tiles * 256;
      if (n1_tiles >= 1 && N1 < N && N - N1 <= 2048) {       // only a SMALL tail is worth a second launch
        VtGemmPlan rest;
        const double c = (double)whole + vt_gemm_plan_cost(M, N - (int)N1, K, epi, nf, &rest, false) + 0.02;
        if (c < best - 0.05) {
          best = c;
          *plan = VtGemmPlan{VT_GEMM_CFG_256x256_W4, 0, (int)N1};
        }
      }
    }
  }
  return best;
}

static VtGemmPlan vt_gemm_plan(int M, int N, int K, int epi, const VtGemmNormFuse* nf) {
  VtGemmPlan plan;
  vt_gemm_plan_cost(M, N, K, epi, nf, &plan, true);
  return plan;
}

int vt_gemm_plan_describe(int M, int N, int K, int epi, int* cfg, int* rows_first, int* cols_first) {
  VT_REQUIRE(M > 0 && N > 0 && K > 0 && cfg && rows_first, "vt_gemm_plan_query: bad arguments");
  if (cols_first) *cols_first = 0;
  if (M <= 64 || (K % 64) != 0) {   // the weight-streaming kernels / the ragged-K fallback, not a tile configuration
    *cfg = VT_GEMM_CFG_SKINNY;
    *rows_first = 0;
    return VT_OK;
  }
  const VtGemmPlan plan = vt_gemm_plan(M, N, K, epi, nullptr);
  *cfg = plan.cfg;
  *rows_first = plan.M1;
  if (cols_first) *cols_first = plan.N1;
  return VT_OK;
}

// measured on MI355X (tools/splitk_bench.py): see the table in DESIGN.md 3.1
// (1088x4096x11008: 149 -> 108 us at ksplit 3; 1024x4096x11008: 114 -> 91 at 4; 300x4096x11008: 96 -> 56 at 8; 577x1024x4096:
// 39 -> 29 at 8; K = 4096 with ksplit < 8 and everything at K = 1024: no gain, the reduce pass eats it)
bool vt_gemm_splitk_pays(int /*M*/, int /*N*/, int K, int ksplit) { return K >= 8192 || (K >= 4096 && ksplit >= 8); }

bool vt_gemm_norm_fold_supported(int M, int N, int K) { return M > 64 && (N % 32) == 0 && (K % 64) == 0; }

int vt_gemm_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, const float* bias,
                   int M, int N, int K, int epi, int cfg, hipStream_t s, const VtGemmNormFuse* nf) {
  VT_REQUIRE(A && W && C, "vt_gemm: null pointer");
  VT_REQUIRE(M > 0 && N > 0 && K > 0, "vt_gemm: empty problem M=%d N=%d K=%d", M, N, K);
  VT_REQUIRE((N % 4) == 0, "vt_gemm: N=%d must be a multiple of 4", N);
  VT_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0, "vt_gemm: lda/ldw must be multiples of 8 elements");
  VT_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)C % 16) == 0,
             "vt_gemm: A/W/C must be 16-byte aligned");
  VT_REQUIRE(ldc % 4 == 0, "vt_gemm: ldc must be a multiple of 4");
  if (nf && !nf->row_scale && !nf->out_partials) nf = nullptr;
  if (nf) {   // folded RMSNorm on the MFMA tile kernels (vt_kernels.h)
    const bool tile_cfg = cfg == VT_GEMM_CFG_AUTO || cfg == VT_GEMM_CFG_256x256_P4 || cfg == VT_GEMM_CFG_256x256_P8 ||
                          cfg == VT_GEMM_CFG_128x128 || cfg == VT_GEMM_CFG_256x128 || cfg == VT_GEMM_CFG_256x256 || cfg == VT_GEMM_CFG_64x128;
    VT_REQUIRE((N % 32) == 0 && (K % 64) == 0 && tile_cfg,
               "vt_gemm: row scale / norm fold needs N %% 32 == 0, K %% 64 == 0 and an MFMA tile configuration (N=%d K=%d cfg=%d)", N, K, cfg);
    if (nf->out_partials)
      VT_REQUIRE(epi == VT_EPI_F32_RESID && nf->out_w && nf->out_xw && nf->out_np >= N / 32 && nf->out_ldp >= M && (nf->ld_xw % 4) == 0,
                 "vt_gemm: norm-fold producer needs the residual epilogue, out_w, out_xw and out_np >= N/32");
  }
  GemmP p{A, W, C, bias, M, N, K, lda, ldw, ldc, nf ? *nf : VtGemmNormFuse{}};
  // (a row range that carries the tile kernels' norm fold stays on the tile kernels whatever its height: remainders of an M-split)
  const bool skinny_path = !nf && (M <= 16 || (M <= 32 && (K % 64) == 0 && cfg != VT_GEMM_CFG_SKINNY_REG)) &&
                           (cfg == VT_GEMM_CFG_AUTO || cfg == VT_GEMM_CFG_SKINNY || cfg == VT_GEMM_CFG_SKINNY_REG);
  // 17..32 rows take the 32-row weight-streaming kernel (weights streamed once). 33..64 rows (short follow-up prompts, larger
  // decode batches): a tile grid is far too small for 256 CUs at N = 4096 (64x128 tiles: 32 workgroups, 0.6 TB/s of weight
  // stream), so the rows go through that kernel in two groups of 32 -- the weights are streamed twice (the repeat mostly from
  // the Infinity Cache), still faster than the tile grid up to N = 16384 (tools/skinny_bench.py).
  if (!nf && cfg == VT_GEMM_CFG_AUTO && M > 32 && M <= 64 && (K % 64) == 0 && N <= 16384) {
    const size_t esz = (epi == VT_EPI_F32 || epi == VT_EPI_F32_RESID) ? 4 : 2;
    for (int m0 = 0; m0 < M; m0 += 32)
      VT_TRY(vt_gemm_launch(A + (size_t)m0 * lda, lda, W, ldw, (char*)C + (size_t)m0 * ldc * esz, ldc, bias, std::min(32, M - m0), N, K,
                            epi, VT_GEMM_CFG_SKINNY, s));
    return VT_OK;
  }
  if (!skinny_path) {
    VT_REQUIRE((K % 64) == 0, "vt_gemm(tile): K=%d must be a multiple of 64", K);
    if (cfg == VT_GEMM_CFG_AUTO || cfg == VT_GEMM_CFG_SKINNY || cfg == VT_GEMM_CFG_SKINNY_REG) {
      // Wave quantisation: when 256-row tiles would leave most of the last round of CUs idle (M = 5120, N = 4096: 320 tiles = 1.25
      // rounds) the planner switches to 320-row tiles, or runs the rows that fill WHOLE rounds on the big tiles and plans the
      // remaining rows again (small tiles fill the CUs at any size). Both launches are plain row ranges of the same GEMM.
      const VtGemmPlan plan = vt_gemm_plan(M, N, K, epi, nf);
      cfg = plan.cfg;
      if (plan.N1 > 0) {   // column split: whole rounds of big tiles over the first N1 columns, the tail columns planned again
        const size_t esz = (epi == VT_EPI_F32 || epi == VT_EPI_F32_RESID) ? 4 : 2;
        const size_t c_off = (epi == VT_EPI_SWIGLU_BF16 ? plan.N1 / 2 : plan.N1) * esz;     // SwiGLU: one output column per gate/up pair
        VT_TRY(vt_gemm_launch(A, lda, W, ldw, C, ldc, bias, M, plan.N1, K, epi, plan.cfg, s, nullptr));
        return vt_gemm_launch(A, lda, W + (size_t)plan.N1 * ldw, ldw, (char*)C + c_off, ldc, bias ? bias + plan.N1 : nullptr, M, N - plan.N1, K,
                              epi, VT_GEMM_CFG_AUTO, s, nullptr);
      }
      if (plan.M1 > 0) {
        const size_t esz = (epi == VT_EPI_F32 || epi == VT_EPI_F32_RESID) ? 4 : 2;
        VT_TRY(vt_gemm_launch(A, lda, W, ldw, C, ldc, bias, plan.M1, N, K, epi, plan.cfg, s, nf));
        const VtGemmNormFuse rest = nf ? vt_nf_rows(*nf, plan.M1) : VtGemmNormFuse{};
        return vt_gemm_launch(A + (size_t)plan.M1 * lda, lda, W, ldw, (char*)C + (size_t)plan.M1 * ldc * esz, ldc, bias, M - plan.M1, N, K,
                              epi, VT_GEMM_CFG_AUTO, s, nf ? &rest : nullptr);
      }
    }
  }
  // algorithmic work: 2*M*N*K FLOP for the MFMA tile kernel; weight bytes for the weight-streaming kernel
  VtProfScope prof(skinny_path ? VT_PROF_GEMM_SKINNY : VT_PROF_GEMM_TILE,
                   skinny_path ? 2.0 * (double)N * (double)K : 2.0 * (double)M * (double)N * (double)K, s);
  if (skinny_path) {
    const bool fma = cfg == VT_GEMM_CFG_SKINNY_REG;   // register-operand kernel on request; it is also the ragged-K fallback
    VT_REQUIRE((K % 8) == 0, "vt_gemm(skinny): K=%d must be a multiple of 8", K);
    switch (epi) {
      case VT_EPI_BF16: return launch_skinny<VT_EPI_BF16>(p, s, fma);
      case VT_EPI_BF16_GELU: return launch_skinny<VT_EPI_BF16_GELU>(p, s, fma);
      case VT_EPI_BF16_QGELU: return launch_skinny<VT_EPI_BF16_QGELU>(p, s, fma);
      case VT_EPI_BF16_RELU: return launch_skinny<VT_EPI_BF16_RELU>(p, s, fma);
      case VT_EPI_F32_RESID: return launch_skinny<VT_EPI_F32_RESID>(p, s, fma);
      case VT_EPI_F32: return launch_skinny<VT_EPI_F32>(p, s, fma);
      case VT_EPI_SWIGLU_BF16:
        VT_REQUIRE((N % 32) == 0, "vt_gemm(swiglu): N must be a multiple of 32");
        return launch_skinny<VT_EPI_SWIGLU_BF16>(p, s, fma);
      default: vt_set_error("vt_gemm: unknown epilogue %d", epi); return VT_ERR_ARG;
    }
  }
  VT_REQUIRE((K % 64) == 0, "vt_gemm(tile): K=%d must be a multiple of 64", K);
  if (epi == VT_EPI_SWIGLU_BF16) VT_REQUIRE((N % 32) == 0, "vt_gemm(swiglu): N must be a multiple of 32");
  if (cfg == VT_GEMM_CFG_256x256_P8) return vt_gemm_p8_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, epi, s, nf);
#ifdef VT_ABLATIONS   // timing ablations / main-loop A/B variants (results are garbage for 100..117): test library only
  if (cfg >= 111 && cfg < 118) return vt_gemm_p8_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, ((cfg - 110) << 8) | 0x1000, s);
  if (cfg >= 100 && cfg < 108) return vt_gemm_p8_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, (cfg - 100) << 8, s);
  if (cfg >= 301 && cfg <= 303) return vt_gemm_rp_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, (cfg - 300) << 8, s);
#endif
  if (cfg == VT_GEMM_CFG_256x256_P4) return vt_gemm_p8_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, epi | 0x1000, s, nf);
  if (cfg == VT_GEMM_CFG_256x256_W4) return vt_gemm_p8_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, epi | 0x4000, s, nf);
  if (cfg == VT_GEMM_CFG_320x256_W4) return vt_gemm_p8_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, epi | 0xc000, s, nf);
  if (cfg == VT_GEMM_CFG_224x256_W4) return vt_gemm_p8_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, epi | 0x24000, s, nf);
  if (cfg == VT_GEMM_CFG_160x128_W4) return vt_gemm_p8_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, epi | 0x10000, s, nf);
  if (cfg == VT_GEMM_CFG_256x256_RP) return vt_gemm_rp_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, epi, s);
  switch (epi) {
    case VT_EPI_BF16: return launch_cfg<VT_EPI_BF16>(p, cfg, s);
    case VT_EPI_BF16_GELU: return launch_cfg<VT_EPI_BF16_GELU>(p, cfg, s);
    case VT_EPI_BF16_QGELU: return launch_cfg<VT_EPI_BF16_QGELU>(p, cfg, s);
    case VT_EPI_BF16_RELU: return launch_cfg<VT_EPI_BF16_RELU>(p, cfg, s);
    case VT_EPI_F32_RESID: return launch_cfg<VT_EPI_F32_RESID>(p, cfg, s);
    case VT_EPI_F32: return launch_cfg<VT_EPI_F32>(p, cfg, s);
    case VT_EPI_SWIGLU_BF16: return launch_cfg<VT_EPI_SWIGLU_BF16>(p, cfg, s);
    default: vt_set_error("vt_gemm: unknown epilogue %d", epi); return VT_ERR_ARG;
  }
}

int vt_gemm_skinny_norm_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, int M, int N, int K, int epi,
                               const VtGemmNormFuse& nf, hipStream_t s) {
  VT_REQUIRE(A && W && C, "vt_gemm(norm-fused): null pointer");
  VT_REQUIRE(M > 0 && M <= 16 && (K % 64) == 0 && (N % 32) == 0, "vt_gemm(norm-fused): needs M <= 16, K %% 64 == 0, N %% 32 == 0 (M=%d N=%d K=%d)", M, N, K);
  VT_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0 && ldc % 4 == 0, "vt_gemm(norm-fused): misaligned leading dimensions");
  if (nf.out_partials) VT_REQUIRE(epi == VT_EPI_F32_RESID && nf.out_w && nf.out_xw, "vt_gemm(norm-fused): producer side needs the residual epilogue, weights and the xw buffer");
  if (nf.in_partials)
    VT_REQUIRE(nf.in_n > 0 && nf.in_n <= 512 && (nf.in_n % 64) == 0 && nf.inv_dim > 0.f,
               "vt_gemm(norm-fused): consumer side needs in_n %% 64 == 0, in_n <= 512 (in_n=%d) and inv_dim", nf.in_n);
  GemmP p{A, W, C, nullptr, M, N, K, lda, ldw, ldc, nf};
  VtProfScope prof(VT_PROF_GEMM_SKINNY, 2.0 * (double)N * (double)K, s);
  switch (epi) {
    case VT_EPI_BF16: return launch_skinny<VT_EPI_BF16>(p, s, false);
    case VT_EPI_F32_RESID: return launch_skinny<VT_EPI_F32_RESID>(p, s, false);
    case VT_EPI_F32: return launch_skinny<VT_EPI_F32>(p, s, false);
    case VT_EPI_SWIGLU_BF16: return launch_skinny<VT_EPI_SWIGLU_BF16>(p, s, false);
    default: vt_set_error("vt_gemm(norm-fused): epilogue %d unsupported", epi); return VT_ERR_ARG;
  }
}

// Residual GEMM (C += A W^T + bias) with an optional split-K workspace. ksplit == 0: decide here -- split only when the 256x256
// grid would cover at most half the chip, enough K is left per split and the workspace is large enough; otherwise the plain
// dispatcher. ksplit >= 2 forces the two-pass path (tests, tools/splitk_bench.py).
int vt_gemm_resid_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, float* C, int ldc, const float* bias, int M, int N,
                         int K, int ksplit, float* partials, size_t partial_bytes, hipStream_t s, const VtGemmNormFuse* nf) {
  int ks = ksplit;
  // K splits that fill one round of the chip with 256x256 tiles: floor(256 / tiles), bounded by the K loop and by 8
  const int sk_tiles = cdiv(M, 256) * cdiv(N, 256);
  const int sk_cand = std::min(std::min(256 / std::max(sk_tiles, 1), (K >> 7) / 2), 8);
  const bool sk_room = partials && (N % 4) == 0 && partial_bytes >= (size_t)std::max(sk_cand, 1) * M * N * sizeof(float);
  if (ks == 0 && M > 64 && vt_gemm_p8_supported(M, N, K)) {
    // A 160x128 grid that covers at least half the chip beats the two-pass split-K on a SHORT K loop (768 x 4096 x 4096: 45 vs 47 us at 5
    // splits). On a LONG one (down_proj, K = 11008) with COLD weights -- what every layer of a prefill sees -- the split that fills the chip
    // wins once it has 3 or more parts (round 5, weights rotated over > 600 MB: 512 rows 88 -> 62 us at 8 splits, 640: 112 -> 70 (5),
    // 768: 114 -> 84 (5), 896: 106 -> 83 (4), 1088: 112 -> 97 (3); round 2's 1077 vs 724 TFLOP/s at 1088 rows was measured on warm weights)
    const bool long_k_split = !nf && sk_room && K >= 8192 && sk_cand >= 3 && (long)sk_tiles * sk_cand >= 192;
    const VtGemmPlan plan = vt_gemm_plan(M, N, K, VT_EPI_F32_RESID, nf);
    if (!long_k_split && plan.M1 == 0 && plan.cfg == VT_GEMM_CFG_160x128_W4 && (long)cdiv(M, 160) * cdiv(N, 128) >= 128)
      return vt_gemm_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, VT_EPI_F32_RESID, plan.cfg, s, nf);
  }
  if (ks == 0 && partials && M > 64 && vt_gemm_p8_supported(M, N, K) && (N % 4) == 0) {
    const int tiles = sk_tiles;
    const int cand = sk_cand;
    // (>= 128 work items; a handful of rows x N = 1024 with a long K loop -- one image's fc2: 257 x 1024 x 4096, 8 tiles -- pays from 64:
    //  40 -> 27 us at 8 splits, 514 rows 38 -> 27)
    const long need = (K >= 4096 && cand >= 8 && N <= 1024) ? 64 : 128;
    if (tiles <= 128 && cand >= 2 && (long)tiles * cand >= need && vt_gemm_splitk_pays(M, N, K, cand) &&
        partial_bytes >= (size_t)cand * M * N * sizeof(float))
      ks = cand;
    if (ks == 0) {
      // the planner's M-split (whole rounds of big tiles + remainder), done here so that the remainder can split K
      const VtGemmPlan plan = vt_gemm_plan(M, N, K, VT_EPI_F32_RESID, nf);
      if (plan.M1 > 0) {
        VT_TRY(vt_gemm_launch(A, lda, W, ldw, C, ldc, bias, plan.M1, N, K, VT_EPI_F32_RESID, plan.cfg, s, nf));
        const VtGemmNormFuse rest = nf ? vt_nf_rows(*nf, plan.M1) : VtGemmNormFuse{};
        return vt_gemm_resid_launch(A + (size_t)plan.M1 * lda, lda, W, ldw, C + (size_t)plan.M1 * ldc, ldc, bias, M - plan.M1, N, K, 0, partials,
                                    partial_bytes, s, nf ? &rest : nullptr);
      }
    }
  }
  if (ks < 2) return vt_gemm_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, VT_EPI_F32_RESID, VT_GEMM_CFG_AUTO, s, nf);
  VT_REQUIRE(partials && partial_bytes >= (size_t)ks * M * N * sizeof(float), "vt_gemm(split-K): workspace too small (%zu bytes for ksplit=%d)",
             partial_bytes, ks);
  VtProfScope prof(VT_PROF_GEMM_TILE, 2.0 * (double)M * (double)N * (double)K, s);
  return vt_gemm_p4_splitk_resid_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, ks, partials, s, nf);
}
