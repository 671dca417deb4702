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
// "skinny" kernel: M <= 16 rows (decode steps, region MLP, last-position lm_head): pure weight
// streaming, one wave per group of output columns, fp32 FMA, wave reduction. HBM-bound.
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
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

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
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n0..n0+3], m = ..+(lane&15), n0 = ..+(lane>>4)*4 ----------------
  const int m_lane = bm0 + wm * WTM + (lane & 15);
  const int n_lane = bn0 + wn * WTN + ((lane >> 4) << 2);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m_lane + mi * 16;
    if (m >= p.M) continue;
    if constexpr (EPI == VT_EPI_SWIGLU_BF16) {
      // W rows are interleaved in blocks of 16: [gate 16 | up 16 | gate 16 | up 16 ...]
      bf16_t* crow = (bf16_t*)p.C + (size_t)m * p.ldc;
#pragma unroll
      for (int ni = 0; ni < NI; ni += 2) {
        const int n = n_lane + ni * 16;  // column of the gate fragment in the interleaved space
        if (n >= p.N) continue;
        const int no = ((bn0 + wn * WTN + ni * 16) >> 1) + ((lane >> 4) << 2);
        const f32x4 g = acc[mi][ni], u = acc[mi][ni + 1];
        u32x2 o;
        o.x = pack_bf16x2(silu(g[0]) * u[0], silu(g[1]) * u[1]);
        o.y = pack_bf16x2(silu(g[2]) * u[2], silu(g[3]) * u[3]);
        *(u32x2*)(crow + no) = o;
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n_lane + ni * 16;
        if (n >= p.N) continue;
        f32x4 v = acc[mi][ni];
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
          f32x4 old = *(const f32x4*)c;
          *(f32x4*)c = old + v;
        } else if constexpr (EPI == VT_EPI_F32) {
          *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = v;
        } else {
          u32x2 o;
          o.x = pack_bf16x2(v[0], v[1]);
          o.y = pack_bf16x2(v[2], v[3]);
          *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// skinny GEMM: M <= 16 (decode steps, region MLP, last-position lm_head). Pure weight streaming:
//   * a block of 4 waves owns 16 output columns (4 per wave); the M activation rows are staged once in LDS as fp32
//     (converted on the way in), K split in slabs of KSLAB so any K fits,
//   * every lane streams 16-byte chunks of its wave's 4 weight rows with non-temporal loads, 2 K-chunks unrolled
//     (8 independent 16-B loads in flight per lane), fp32 FMA, one wave reduction per (column, row) at the end.
// Roofline: HBM (2*N*K weight bytes per launch, read exactly once).
// ------------------------------------------------------------------------------------------------
template <int MROWS, int EPI>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmP p) {
  constexpr int NPW = 2;        // output columns per wave
  constexpr int UNR = 4;        // K chunks in flight per column  -> NPW*UNR = 8 independent 16-B loads per lane
  constexpr int KSLAB = 2048;   // activation slab staged in LDS: MROWS * KSLAB * 4 B (<= 128 KiB at MROWS = 16)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = (float*)smem;     // [MROWS][KSLAB]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // plain epilogues: the wave owns columns n0, n0+1. SwiGLU: it owns the (gate, up) pair of output column q -- rows
  // 32*(q/16) + q%16 and +16 of the block-interleaved weight -- so silu(gate)*up is formed in the same lane.
  const int q = blockIdx.x * 4 + wave;
  const int n0 = (EPI == VT_EPI_SWIGLU_BF16) ? ((q >> 4) * 32 + (q & 15)) : q * NPW;
  constexpr int NSTEP = (EPI == VT_EPI_SWIGLU_BF16) ? 16 : 1;
  float acc[NPW][MROWS];
#pragma unroll
  for (int j = 0; j < NPW; ++j)
#pragma unroll
    for (int m = 0; m < MROWS; ++m) acc[j][m] = 0.f;
  const bf16_t* wrow[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) wrow[j] = p.W + (size_t)min(n0 + j * NSTEP, p.N - 1) * p.ldw;

  for (int k0 = 0; k0 < p.K; k0 += KSLAB) {
    const int kw = min(KSLAB, p.K - k0);   // multiple of 8
    __syncthreads();                       // previous slab fully consumed
    for (int i = threadIdx.x; i < MROWS * (kw >> 3); i += 256) {
      const int m = i / (kw >> 3), c = i % (kw >> 3);
      u32x4 x = {0u, 0u, 0u, 0u};
      if (m < p.M) x = *(const u32x4*)(p.A + (size_t)m * p.lda + k0 + c * 8);
      float* d = xs + m * KSLAB + c * 8;
      *(f32x4*)d = (f32x4){bf16lo_to_f32(x[0]), bf16hi_to_f32(x[0]), bf16lo_to_f32(x[1]), bf16hi_to_f32(x[1])};
      *(f32x4*)(d + 4) = (f32x4){bf16lo_to_f32(x[2]), bf16hi_to_f32(x[2]), bf16lo_to_f32(x[3]), bf16hi_to_f32(x[3])};
    }
    __syncthreads();
    const int chunks = kw >> 3;
    for (int c = lane; c < chunks; c += 64 * UNR) {
      u32x4 w[UNR][NPW];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int cu = c + 64 * u;
#pragma unroll
        for (int j = 0; j < NPW; ++j)
          w[u][j] = (cu < chunks) ? __builtin_nontemporal_load((const u32x4*)(wrow[j] + k0 + cu * 8)) : (u32x4){0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int cu = min(c + 64 * u, chunks - 1);   // out-of-range chunks carry zero weights
#pragma unroll
        for (int m = 0; m < MROWS; ++m) {
          const f32x4 xa = *(const f32x4*)(xs + m * KSLAB + cu * 8);
          const f32x4 xb = *(const f32x4*)(xs + m * KSLAB + cu * 8 + 4);
#pragma unroll
          for (int j = 0; j < NPW; ++j) {
            const u32x4 ww = w[u][j];
            float a = acc[j][m];
            a = fmaf(bf16lo_to_f32(ww[0]), xa[0], a);
            a = fmaf(bf16hi_to_f32(ww[0]), xa[1], a);
            a = fmaf(bf16lo_to_f32(ww[1]), xa[2], a);
            a = fmaf(bf16hi_to_f32(ww[1]), xa[3], a);
            a = fmaf(bf16lo_to_f32(ww[2]), xb[0], a);
            a = fmaf(bf16hi_to_f32(ww[2]), xb[1], a);
            a = fmaf(bf16lo_to_f32(ww[3]), xb[2], a);
            a = fmaf(bf16hi_to_f32(ww[3]), xb[3], a);
            acc[j][m] = a;
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NPW; ++j)
#pragma unroll
    for (int m = 0; m < MROWS; ++m) acc[j][m] = wave_sum(acc[j][m]);

  if constexpr (EPI == VT_EPI_SWIGLU_BF16) {
    if (lane == 0 && n0 + 16 < p.N) {
#pragma unroll
      for (int m = 0; m < MROWS; ++m)
        if (m < p.M) ((bf16_t*)p.C)[(size_t)m * p.ldc + q] = f32_to_bf16(silu(acc[0][m]) * acc[1][m]);
    }
    return;
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int n = n0 + j;
      if (n >= p.N) continue;
      const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int m = 0; m < MROWS; ++m) {
        if (m >= p.M) continue;
        float v = acc[j][m] + b;
        if constexpr (EPI == VT_EPI_BF16_GELU) v = gelu_erf(v);
        if constexpr (EPI == VT_EPI_BF16_QGELU) v = quick_gelu(v);
        if constexpr (EPI == VT_EPI_BF16_RELU) v = fmaxf(v, 0.f);
        if constexpr (EPI == VT_EPI_F32_RESID) {
          float* c = (float*)p.C + (size_t)m * p.ldc + n;
          *c = *c + v;
        } else if constexpr (EPI == VT_EPI_F32) {
          ((float*)p.C)[(size_t)m * p.ldc + n] = v;
        } else {
          ((bf16_t*)p.C)[(size_t)m * p.ldc + n] = f32_to_bf16(v);
        }
      }
    }
  }
}

// SwiGLU for the skinny path: gate/up come out of the interleaved weight as separate columns, so the
// skinny kernel writes fp32 [M][N] to scratch and this kernel pairs them.
__global__ void swiglu_pair_kernel(const float* gu, bf16_t* out, int M, int N, int ldo) {
  const int half = N >> 1;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * half) return;
  const int m = idx / half, j = idx % half;
  const int blk = j >> 4, r = j & 15;
  const float g = gu[(size_t)m * N + blk * 32 + r];
  const float u = gu[(size_t)m * N + blk * 32 + 16 + r];
  out[(size_t)m * ldo + j] = f32_to_bf16(silu(g) * u);
}

template <int BM, int BN, int WM, int WN, int EPI>
int launch_tile(const GemmP& p, hipStream_t s) {
  constexpr int smem = 2 * (BM + BN) * 64 * 2;
  static bool attr_done = false;  // benign race: idempotent
  auto kern = gemm_bt_kernel<BM, BN, WM, WN, EPI>;
  if (!attr_done) {
    VT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_done = true;
  }
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

template <int MR, int EPI>
int launch_skinny_m(const GemmP& p, hipStream_t s) {
  constexpr int smem = MR * 2048 * 4;
  auto kern = gemm_skinny_kernel<MR, EPI>;
  static bool done = false;
  if (!done) {
    VT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    done = true;
  }
  const int waves = (EPI == VT_EPI_SWIGLU_BF16) ? p.N / 2 : cdiv(p.N, 2);
  hipLaunchKernelGGL(kern, dim3(cdiv(waves, 4)), dim3(256), smem, s, p);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

template <int EPI>
int launch_skinny(const GemmP& p, hipStream_t s) {
  if (p.M <= 4) return launch_skinny_m<4, EPI>(p, s);
  if (p.M <= 8) return launch_skinny_m<8, EPI>(p, s);
  return launch_skinny_m<16, EPI>(p, s);
}

}  // namespace

int vt_gemm_pick_cfg(int M, int N, int K) {
  // Measured on MI355X (tools/gemm_bench.py, profiles/): the 8-phase 256x256 kernel wins whenever its grid fills
  // whole rounds of the 256 CUs (one 128-KiB-LDS workgroup per CU); when the last round would be mostly empty the
  // 128x128 kernel (two workgroups per CU, 4x more tiles) quantises better. The register-pipelined variant
  // (VT_GEMM_CFG_256x256_RP) is 3-8% faster than the 8-phase kernel on isolated back-to-back launches but measured
  // equal (bf16 epilogue) to 9% slower (fp32 residual epilogue) inside the full prefill step, so it stays opt-in.
  if (M <= 64) return VT_GEMM_CFG_64x128;
  const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
  const long rounds = (t256 + 255) / 256;
  const double waste = 1.0 - (double)t256 / (double)(rounds * 256);
  if (vt_gemm_p8_supported(M, N, K) && waste <= 0.15) return VT_GEMM_CFG_256x256_P8;
  const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
  if (t128 < 256 && M <= 640) return VT_GEMM_CFG_64x128;
  return VT_GEMM_CFG_128x128;
}

int vt_gemm_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, const float* bias,
                   int M, int N, int K, int epi, int cfg, void* skinny_scratch, hipStream_t s) {
  VT_REQUIRE(A && W && C, "vt_gemm: null pointer");
  VT_REQUIRE(M > 0 && N > 0 && K > 0, "vt_gemm: empty problem M=%d N=%d K=%d", M, N, K);
  VT_REQUIRE((N % 4) == 0, "vt_gemm: N=%d must be a multiple of 4", N);
  VT_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0, "vt_gemm: lda/ldw must be multiples of 8 elements");
  VT_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)C % 16) == 0,
             "vt_gemm: A/W/C must be 16-byte aligned");
  VT_REQUIRE(ldc % 4 == 0, "vt_gemm: ldc must be a multiple of 4");
  GemmP p{A, W, C, bias, M, N, K, lda, ldw, ldc};
  const bool skinny_path = (M <= 16) && (cfg == VT_GEMM_CFG_AUTO || cfg == VT_GEMM_CFG_SKINNY);
  if (!skinny_path) {
    VT_REQUIRE((K % 64) == 0, "vt_gemm(tile): K=%d must be a multiple of 64", K);
  if (cfg == VT_GEMM_CFG_AUTO || cfg == VT_GEMM_CFG_SKINNY) {
      cfg = vt_gemm_pick_cfg(M, N, K);
      // Wave quantisation: when the 256x256 grid would leave most of its last round of CUs idle (e.g. M=5120, N=4096:
      // 320 tiles = 1.25 rounds), run the rows that fill WHOLE rounds on the 256x256 kernel and the remaining rows on the
      // small-tile kernel (4x more, 4x shorter tiles fill the CUs again). Both launches are plain row ranges of the same GEMM.
      if (cfg != VT_GEMM_CFG_256x256_P8 && vt_gemm_p8_supported(M, N, K) && K >= 2048 && M > 256) {
        const int tiles_n = cdiv(N, 256);
        int g = tiles_n, b = 256;
        while (b) { const int t = g % b; g = b; b = t; }   // gcd(tiles_n, 256)
        const long unit = 256L * (256 / g);                 // rows per whole round
        const long M1 = (M / unit) * unit;
        if (M1 >= unit && M1 < M) {
          const size_t esz = (epi == VT_EPI_F32 || epi == VT_EPI_F32_RESID) ? 4 : 2;
          VT_TRY(vt_gemm_launch(A, lda, W, ldw, C, ldc, bias, (int)M1, N, K, epi, VT_GEMM_CFG_256x256_P8, skinny_scratch, s));
          return vt_gemm_launch(A + (size_t)M1 * lda, lda, W, ldw, (char*)C + (size_t)M1 * ldc * esz, ldc, bias, M - (int)M1, N, K,
                                epi, VT_GEMM_CFG_AUTO, skinny_scratch, s);
        }
      }
    }
  }
  // algorithmic work: 2*M*N*K FLOP for the MFMA tile kernel; weight bytes for the weight-streaming kernel
  VtProfScope prof(skinny_path ? VT_PROF_GEMM_SKINNY : VT_PROF_GEMM_TILE,
                   skinny_path ? 2.0 * (double)N * (double)K : 2.0 * (double)M * (double)N * (double)K, s);
  const bool skinny = (M <= 16) && (cfg == VT_GEMM_CFG_AUTO || cfg == VT_GEMM_CFG_SKINNY);
  if (skinny) {
    VT_REQUIRE((K % 8) == 0, "vt_gemm(skinny): K=%d must be a multiple of 8", K);
    switch (epi) {
      case VT_EPI_BF16: return launch_skinny<VT_EPI_BF16>(p, s);
      case VT_EPI_BF16_GELU: return launch_skinny<VT_EPI_BF16_GELU>(p, s);
      case VT_EPI_BF16_QGELU: return launch_skinny<VT_EPI_BF16_QGELU>(p, s);
      case VT_EPI_BF16_RELU: return launch_skinny<VT_EPI_BF16_RELU>(p, s);
      case VT_EPI_F32_RESID: return launch_skinny<VT_EPI_F32_RESID>(p, s);
      case VT_EPI_F32: return launch_skinny<VT_EPI_F32>(p, s);
      case VT_EPI_SWIGLU_BF16:
        VT_REQUIRE((N % 32) == 0, "vt_gemm(swiglu): N must be a multiple of 32");
        return launch_skinny<VT_EPI_SWIGLU_BF16>(p, s);
      default: vt_set_error("vt_gemm: unknown epilogue %d", epi); return VT_ERR_ARG;
    }
  }
  VT_REQUIRE((K % 64) == 0, "vt_gemm(tile): K=%d must be a multiple of 64", K);
  if (epi == VT_EPI_SWIGLU_BF16) VT_REQUIRE((N % 32) == 0, "vt_gemm(swiglu): N must be a multiple of 32");
  if (cfg == VT_GEMM_CFG_256x256_P8) return vt_gemm_p8_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, epi, s);
  if (cfg >= 100 && cfg < 108) return vt_gemm_p8_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, (cfg - 100) << 8, s);
  if (cfg >= 301 && cfg <= 303) return vt_gemm_rp_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, (cfg - 300) << 8, s);
  if (cfg >= 200 && cfg < 208) return vt_gemm_w4_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, (cfg - 200) << 8, s);
  if (cfg == VT_GEMM_CFG_256x256_RP) return vt_gemm_rp_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, epi, s);
  if (cfg == VT_GEMM_CFG_256x256_W4) return vt_gemm_w4_launch(A, lda, W, ldw, C, ldc, bias, M, N, K, epi, s);
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
