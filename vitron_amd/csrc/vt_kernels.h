// vt_kernels.h -- internal launch prototypes shared by the .hip translation units and vt_api.hip.
// Not part of the public C-ABI (that is include/vitron_hip.h).
#pragma once
#include <vector>
#include "vt_common.h"
#include "../../include/vitron_hip.h"

struct VtAttnSeq {  // device-side view of one row of seq_desc (int32 x 4)
  int q_row0;       // first row of this sequence's queries in the packed Q matrix
  int q_len;        // number of query rows
  int kv_len;       // number of valid keys (past + q_len)
  int table_off;    // offset into the tile table of this sequence's first 64-key tile
};

// ---- vt_gemm.hip ----------------------------------------------------------------------------------
int vt_gemm_pick_cfg(int M, int N, int K);
int vt_gemm_plan_describe(int M, int N, int K, int epi, int* cfg, int* rows_first, int* cols_first = nullptr);   // vt_gemm.hip: the planner's verdict, no launch
struct VtGemmNormFuse;
int vt_gemm_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, const float* bias,
                   int M, int N, int K, int epi, int cfg, hipStream_t s,
                   const VtGemmNormFuse* nf = nullptr);
// RMSNorm folded into the M <= 16 weight-streaming GEMMs of a decode step (no separate norm launches):
//   producer side (VT_EPI_F32_RESID): after x += A W^T the epilogue also stores out_xw = bf16(x * out_w[n]) and, per row and
//   16-column block, the partial sum of x^2 (out_partials[m][N/16]);
//   consumer side: A = that xw buffer; the epilogue scales row m by rsqrt(sum(in_partials[m][0..in_n)) * inv_dim + eps) before
//   the activation -- W (w .* x) * rstd == W (x * rstd .* w), the per-row scalar commutes with the contraction.
// The MFMA tile kernels (rows > 64: prefill) use the same fold with two differences: the partial sums are per 32-column
//   group (out_partials[group][m], out_np = N/32 groups of out_ldp rows), and the consumer takes the finished per-row factor (row_scale[m] =
//   rstd, produced from the partials by vt_rowscale_finalize_launch) instead of summing partials in its epilogue.
struct VtGemmNormFuse {
  const float* in_partials = nullptr;
  int in_n = 0;
  float inv_dim = 0.f, eps = 0.f;
  const float* out_w = nullptr;
  bf16_t* out_xw = nullptr;
  int ld_xw = 0;
  float* out_partials = nullptr;
  int out_np = 0;                      // tile kernels: 32-column groups per row; out_partials is [out_np][out_ldp] (group-major:
                                       // the 16 rows a store instruction covers are 64 contiguous bytes)
  int out_ldp = 0;                     // tile kernels: row capacity of one group's array
  const float* row_scale = nullptr;    // tile kernels, consumer side: accumulator of row m is scaled by row_scale[m]
};
// the same descriptor for the row range [m0, ...) of the problem (M-split, remainders)
inline VtGemmNormFuse vt_nf_rows(const VtGemmNormFuse& nf, long m0) {
  VtGemmNormFuse r = nf;
  if (r.in_partials) r.in_partials += m0 * r.in_n;
  if (r.out_xw) r.out_xw += m0 * r.ld_xw;
  if (r.out_partials) r.out_partials += (r.out_ldp ? m0 : m0 * r.out_np);
  if (r.row_scale) r.row_scale += m0;
  return r;
}
int vt_gemm_skinny_norm_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, int M, int N, int K,
                               int epi, const VtGemmNormFuse& nf, hipStream_t s);

// ---- fused QKV epilogue (vt_gemm8.hip) ---------------------------------------------------------------------
// The QKV projection of a prefill writes its result where the attention kernels read it: q rotated in place in the fused-QKV
// buffer, k rotated into the K pages, v transposed into the V^T pages -- what vt_kv_tiles does as a separate pass over 252 MB
// per layer. Internal epilogue id (not part of the public enum); head_dim 128, rotary on.
#define VT_EPI_QKV_PAGES 7
struct VtQkvFuse {
  bf16_t* k_pages = nullptr;        // this layer's K pages  [page][head][64][128]
  bf16_t* vt_pages = nullptr;       // this layer's V^T pages [page][head][128][64]
  const int* row_slot = nullptr;    // [M]: (tail << 26) | (page * 64 + slot); tail = slots to zero behind the LAST new row of a sequence
  const int* positions = nullptr;   // [M] rotary position of every row
  const float* rope_cos = nullptr;  // [rope_len][64]
  const float* rope_sin = nullptr;
  int heads = 0;                    // heads per q / k / v section (H = heads * 128)
};
// row -> cache slot of every new row (one launch per decoder pass)
int vt_row_slot_launch(const VtAttnSeq* seqs, int nseq, int max_q_len, const int* tile_table, int* row_slot, hipStream_t s);
// true when the QKV GEMM of (rows x 3H x H) runs as ONE launch of the 256x256 ping-pong kernel, i.e. can carry the fused epilogue
bool vt_gemm_qkv_fused_supported(int rows, int H, int head_dim);
int vt_gemm_qkv_fused_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, bf16_t* qkv, int ldqkv, int M, int H,
                             const VtQkvFuse& qf, hipStream_t s);

// ---- vt_gemm8.hip (256x256 tile, 8-phase pipeline) ---------------------------------------------------
bool vt_gemm_p8_supported(int M, int N, int K);
bool vt_gemm_splitk_pays(int M, int N, int K, int ksplit);
int vt_gemm_resid_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, float* C, int ldc, const float* bias, int M, int N,
                         int K, int ksplit, float* partials, size_t partial_bytes, hipStream_t s,
                         const VtGemmNormFuse* nf = nullptr);
int vt_gemm_p4_splitk_resid_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, float* C, int ldc, const float* bias, int M,
                                   int N, int K, int ksplit, float* partials, hipStream_t s, const VtGemmNormFuse* nf = nullptr);
int vt_gemm_p8_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, const float* bias, int M,
                      int N, int K, int epi, hipStream_t s, const VtGemmNormFuse* nf = nullptr);
// tile-kernel norm fold: is (M, N, K) a shape whose residual GEMM can produce / whose GEMM can consume the folded norm?
bool vt_gemm_norm_fold_supported(int M, int N, int K);

int vt_gemm_rp_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, const float* bias, int M,
                      int N, int K, int epi, hipStream_t s);

// ---- vt_mx4.hip / vt_gemm8x.inc: precise level 3 (16-bit product + MX-FP4 product of the rounding remainder, one accumulator) -------------
// W4 [N][K/2] + wexp[N] (one e8m0 exponent per output feature); A4 [M][K/2] + aexp (one exponent per row and 32 k, in the order the
// GEMM's lanes fetch them; vt_mx4_aexp_bytes(M, K) bytes -- include/vitron_hip.h -- rows padded to whole 256-row tiles)
int vt_mx4_quant_weights_launch(const bf16_t* W, int ldw, int N, int K, uint8_t* W4, uint8_t* wexp, hipStream_t s);
int vt_mx4_quant_lo_launch(const bf16_t* lo, int ld, int M, int K, uint8_t* A4, uint8_t* aexp, hipStream_t s);
int vt_rmsnorm_mx_launch(const float* x, const int* idx, const float* w, bf16_t* y, uint8_t* A4, uint8_t* aexp, int rows, int D, float eps,
                         hipStream_t s);
int vt_gemm_mx_launch(const bf16_t* A, int lda, const uint8_t* A4, const uint8_t* aexp, const bf16_t* W, int ldw, const uint8_t* W4,
                      const uint8_t* wexp, void* C, int ldc, const float* bias, int M, int N, int K, int epi, int ksplit, size_t slab,
                      uint8_t* out4, uint8_t* oexp, hipStream_t s);
#define VT_EPI_SWIGLU_MX 8   // internal epilogues of vt_gemm_mx_launch: an activation with the level 3 operand out (C = op16(v), out4 / oexp = image of
#define VT_EPI_GELU_MX 9     // v - f32(C)): SwiGLU over the interleaved gate / up columns; erf-GELU / quick-GELU of acc + bias (the towers' fc1)
#define VT_EPI_QGELU_MX 10
int vt_gemm_mx_resid_launch(const bf16_t* A, int lda, const uint8_t* A4, const uint8_t* aexp, const bf16_t* W, int ldw, const uint8_t* W4,
                            const uint8_t* wexp, float* C, int ldc, const float* bias, int M, int N, int K, float* partials, size_t partial_bytes,
                            hipStream_t s);
int vt_layernorm_mx_launch(const float* x, const float* gamma, const float* beta, bf16_t* y, uint8_t* A4, uint8_t* aexp, int rows, int D,
                           float eps, hipStream_t s);

// ---- vt_norm.hip ----------------------------------------------------------------------------------
int vt_layernorm_launch(float* x, const float* temb, int T, int tokens_per_frame, const float* gamma,
                        const float* beta, bf16_t* y, int rows, int D, float eps, hipStream_t s);
int vt_rmsnorm_launch(const float* x, const int* idx, const float* w, bf16_t* y, int rows, int D, float eps,
                      hipStream_t s);
// precise modes (operand pairs: hi = op(v), lo = op(v - f32(hi))). RMSNorm / LayerNorm with a pair output; an activation (0 erf-GELU,
// 1 quick-GELU, 2 SwiGLU on 16-interleaved gate/up columns) over an fp32 GEMM output with a pair output; fp32 rows -> pair (optionally
// dropping the CLS row of every frame); fp32 += f32(op16)
int vt_rmsnorm_hilo_launch(const float* x, const int* idx, const float* w, bf16_t* y, bf16_t* ylo, int rows, int D, float eps, hipStream_t s);
int vt_layernorm_hilo_launch(const float* x, const float* gamma, const float* beta, bf16_t* y, bf16_t* ylo, int rows, int D, float eps,
                             hipStream_t s);
int vt_act_pair_launch(const float* in, int ldin, bf16_t* hi, bf16_t* lo, int rows, int Nout, int mode, hipStream_t s);
int vt_f32_to_pair_launch(const float* x, bf16_t* hi, bf16_t* lo, size_t out_rows, int G2, int D, hipStream_t s);
int vt_add_op_to_f32_launch(float* dst, const bf16_t* a, size_t n, hipStream_t s);
int vt_rowscale_finalize_launch(const float* partials, int np, int ldp, int rows, float inv_dim, float eps, float* out, hipStream_t s);
int vt_gather_f32_to_bf16_launch(const float* in, const int* idx, bf16_t* out, int rows, int D, hipStream_t s);
int vt_bf16_to_f32_launch(const bf16_t* in, float* out, size_t n, hipStream_t s);
int vt_add_f32_launch(float* dst, const float* a, size_t n, hipStream_t s);
// drop the CLS row of every frame: out_bf16[f*G2+p] = bf16(x[f*(G2+1)+1+p])
int vt_drop_cls_launch(const float* x, bf16_t* out, int F, int G2, int D, hipStream_t s);

// ---- vt_vit.hip -----------------------------------------------------------------------------------
int vt_im2col_launch(const void* pixels, int pix_dtype, bf16_t* patches, int B, int T, int H, int W, int P,
                     int Kpad, int video_layout, hipStream_t s);
int vt_vit_embed_launch(const float* patch_out, const float* cls, const float* pos, const float* g,
                        const float* b, float* x, int F, int G2, int D, float eps, hipStream_t s);
// seq_desc / tile table of the ViT's spatial attention: frame f = {f*N, N, N, f*ntiles}, table[i] = i
int vt_vit_attn_meta_launch(int* seq_desc, int* tile_table, int F, int N, hipStream_t s);

// ---- vt_attn.hip ----------------------------------------------------------------------------------
int vt_flash_attn_launch(const bf16_t* Q, int ldq, const bf16_t* Kt, const bf16_t* Vt, const int* tile_table,
                         const VtAttnSeq* seqs, int nseq, int max_q_len, bf16_t* O, int ldo, int heads, int HD,
                         int causal, float scale, hipStream_t s, uint8_t* O4 = nullptr, uint8_t* oexp = nullptr);   // O4 / oexp: level 3 operand out (vt_mx4.h)
// vt_attn_w4.hip: the one-wave-per-SIMD prefill kernel (head_dim 128, 256-row blocks). placed = 1: hand-placed sub-iterations,
// 0: the same pipeline with its stages run one after the other (reference of the placed schedule)
bool vt_flash_attn_w4_supported(int HD, int max_q_len, int heads, int nseq);
std::vector<int> vt_flash_attn_w4_block_order(int heads, int nqb, int nseq, int ncu);
int vt_flash_attn_w4_launch(const bf16_t* Q, int ldq, const bf16_t* Kt, const bf16_t* Vt, const int* tile_table, const VtAttnSeq* seqs,
                            int nseq, int max_q_len, bf16_t* O, int ldo, int heads, int causal, float scale_log2e, int placed,
                            hipStream_t s, uint8_t* O4 = nullptr, uint8_t* oexp = nullptr);
// which prefill kernel vt_flash_attn_launch runs for head_dim 128 (vt_flash_attn_select): 0 auto, 1 two-waves-per-SIMD kernel,
// 2 one-wave-per-SIMD kernel (placed), 3 the same unplaced
extern int g_vt_flash_attn_kernel;
extern int g_vt_flash_attn_order;  // 1 (default): causal launches of the one-wave-per-SIMD kernel use the planned dispatch order
extern int g_vt_flash_attn_wgs;   // persistent form: workgroup cap (0 = one per CU); tests use it to force many blocks per workgroup
int vt_kv_tiles_launch(bf16_t* qkv, int ldqkv, int q_col0, int k_col0, int v_col0, bf16_t* Kt, bf16_t* Vt,
                       const int* tile_table, const VtAttnSeq* seqs, int nseq, int max_new_tiles, int heads, int HD,
                       const float* rope_cos, const float* rope_sin, const int* positions, hipStream_t s);
// precise_qk (vt_llama_model.precise_qk): fp32 q / k in, rotary in fp32, q and k written as operand pairs (hi + lo); the matching
// attention kernel computes K_hi.(Q_hi + Q_lo)^T + K_lo.Q_hi^T
int vt_kv_tiles_precise_launch(const float* qk32, int ld32, int v_f32, bf16_t* qkv, int ldqkv, int q_col0, int v_col0, bf16_t* qlo, bf16_t* Kt, bf16_t* Vt,
                               bf16_t* klo, const int* tile_table, const VtAttnSeq* seqs, int nseq, int max_new_tiles, int heads, int HD,
                               const float* rope_cos, const float* rope_sin, const int* positions, hipStream_t s);
int vt_flash_attn_precise_launch(const bf16_t* Q, int ldq, const bf16_t* Qlo, int ldqlo, const bf16_t* Kt, const bf16_t* Klo,
                                 int klo_tiles_per_seq, const bf16_t* Vt, const int* tile_table, const VtAttnSeq* seqs, int nseq,
                                 int max_q_len, bf16_t* O, int ldo, bf16_t* Olo, int heads, int HD, int causal, float scale, hipStream_t s);
size_t vt_attn_decode_scratch_bytes(int nseq, int heads, int HD, int max_kv_len);
int vt_attn_decode_fused_launch(const bf16_t* qkv, int ldqkv, int q_col0, int k_col0, int v_col0, bf16_t* Kt, bf16_t* Vt,
                                const int* tile_table, const VtAttnSeq* seqs, int nseq, bf16_t* O, int ldo, int heads, int HD,
                                float scale, const float* rope_cos, const float* rope_sin, const int* positions, hipStream_t s);
int vt_attn_decode_launch(const bf16_t* Q, int ldq, const bf16_t* Kt, const bf16_t* Vt, const int* tile_table,
                          const VtAttnSeq* seqs, int nseq, bf16_t* O, int ldo, int heads, int HD, float scale,
                          int max_kv_len, float* scratch, size_t scratch_bytes, hipStream_t s);
int vt_attn_temporal_launch(const bf16_t* qkv, bf16_t* out, int B, int T, int N, int heads, hipStream_t s);
// precise level 2 of the towers: the temporal attention on fp32 q | k | v rows (row stride ld), output as an operand pair
int vt_attn_temporal_f32_launch(const float* qkv, int ld, bf16_t* out, bf16_t* out_lo, int B, int T, int N, int heads, hipStream_t s);

// ---- vt_region.hip --------------------------------------------------------------------------------
// cell mask / count / masked mean per (box, 64-channel slab); optionally also LocationEncoder layer 0:
// loc_out[b][n] = op16(relu(coords[b][0..4) . loc_w0[n][0..4) + loc_b0[n])), n < loc_n; coords fp32 [B][4]
int vt_region_pool_launch(const bf16_t* feats, const int* slices, int B, int G, int image_size, int D,
                          bf16_t* pooled, int* cell_mask, int* cell_count, const float* coords, const bf16_t* loc_w0,
                          const float* loc_b0, int loc_n, bf16_t* loc_out, int ld_loc, hipStream_t s);

// ---- vt_llama.hip ---------------------------------------------------------------------------------
int vt_embed_splice_launch(const bf16_t* tok_table, int vocab, const bf16_t* vis, int vis_rows, const bf16_t* reg, int reg_rows,
                           const int* plan, int rows, int H, bf16_t* out, hipStream_t s);
int vt_argmax_launch(const float* logits, int rows, int V, int ldl, int* out_ids, hipStream_t s);
int vt_decode_feed_launch(const bf16_t* tok_table, int H, int vocab, const int* next_ids, int* finished, const int* eos_ids,
                          int n_eos, int pad_id, int* tokens_out, bf16_t* x, int* seq_desc, int* positions, int nseq,
                          hipStream_t s);
int vt_cross_entropy_launch(const float* logits, int rows, int V, int ldl, const int* labels, int ignore_index, float* row_nll,
                            float* loss, hipStream_t s);
int vt_sample_top_p_launch(const float* logits, int rows, int V, int ldl, float temperature, int top_k, float top_p, uint64_t seed,
                           uint64_t step, int* out_ids, int* kept_count, hipStream_t s);

// ---- vt_preproc.hip -------------------------------------------------------------------------------
int vt_preprocess_launch(const void* src, int src_u8, int hwc, int F, int H, int W, int bicubic, int S, const float* mean,
                         const float* std, int flip, void* dst, int dst_dtype, long dst_sc, long dst_sf, hipStream_t s);

// ---- profiling (vt_api.hip) -----------------------------------------------------------------------------
// RAII bracket: records start/stop events on `s` around a launch when profiling is enabled.
bool vt_prof_enabled();
void vt_prof_start(int cls, double work, hipStream_t s);
void vt_prof_stop(hipStream_t s);
struct VtProfScope {
  hipStream_t s;
  bool on;
  VtProfScope(int cls, double work, hipStream_t st) : s(st), on(vt_prof_enabled()) {
    if (on) vt_prof_start(cls, work, s);
  }
  ~VtProfScope() {
    if (on) vt_prof_stop(s);
  }
};
