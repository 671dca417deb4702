/* vitron_hip.h -- C ABI of libvitron_hip.so: the MI355X (gfx950) implementation of Vitron's multimodal
 * forward pass (LanguageBind ViT image/video tower -> region_extractor -> mm_projector -> Vicuna/LLaMA decoder).
 *
 * The reference (SkyworkAI/Vitron) has no FFI on this path: the boundary is the Python surface of
 * vitron.model. This ABI sits directly below that surface; vitron_amd/ binds it with
 * ctypes and keeps the reference's Python signatures. Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - plain C, no torch types; every tensor is a raw DEVICE pointer owned by the caller (PyTorch's caching
 *     allocator in practice) and valid for the duration of the call. The library never allocates or frees
 *     tensor memory: scratch comes in as (workspace, workspace_bytes); query the size with *_workspace_bytes.
 *   - 16-bit tensors are uint16_t* (raw bits), row-major. Linear weights keep torch's [out][in] layout.
 *   - OPERAND FORMAT. The same sources are built twice with the same ABI: libvitron_hip.so computes with bf16 operands
 *     (BASELINE.json's dtype) and libvitron_hip_f16.so with IEEE fp16 operands -- the reference's own inference dtype
 *     (vitron/model/builder.py:47 torch_dtype=float16; towers :153,161). vt_operand_format() says which one a handle is.
 *     Wherever this header says "bf16" for a tensor (weights, activations, embeddings, K pages, pixel input, outputs) read
 *     "the library's operand format"; the historical names (vt_gemm_bf16, VT_EPI_BF16, VT_DTYPE_BF16) are kept for both.
 *     Everything else is identical in the two libraries: fp32 residual stream / accumulation / softmax statistics / biases,
 *     fp16 V^T pages and softmax weights. The fp16 library saturates operand stores at +-65504 (the reference would
 *     produce inf); bf16 weights of magnitude below 2^-14 lose mantissa bits when repacked to fp16 (absolute error <= 2^-25).
 *   - every function is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *     re-entrant, and returns 0 on success or a negative vt_status; vt_last_error() gives the message of the
 *     calling thread's last failure. Nothing throws, nothing synchronises the device.
 *   - ONE exception to "never allocates": the head_dim-128 prefill attention keeps, per (device, heads, query blocks, sequences)
 *     shape, its planned block-dispatch order in a small device buffer (heads * blocks * sequences int32) that it allocates the
 *     first time the shape is launched and uploads with hipMemcpyAsync on `stream` (no device synchronisation; other streams are
 *     ordered behind the upload by an event). A HIP graph capture must see a shape for the second time; vt_flash_attn_select(5)
 *     turns the planned order -- and with it the allocation -- off. All cached per-process state is keyed by device ordinal.
 *   - "host struct" arguments (vt_vit_model, vt_llama_model, ...) are read on the host during the call; the
 *     pointers inside them are device pointers.
 */
#ifndef VITRON_HIP_H
#define VITRON_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { VT_STATUS_OK = 0, VT_STATUS_ARG = -1, VT_STATUS_HIP = -2, VT_STATUS_WORKSPACE = -3 } vt_status;

/* GEMM epilogues: C = epi(A.W^T + bias) */
enum {
  VT_EPI_BF16 = 0,        /* bf16 out                                                                     */
  VT_EPI_BF16_GELU = 1,   /* bf16 out, exact-erf GELU   (CLIPMLP 'gelu', mm_projector nn.GELU)               */
  VT_EPI_BF16_QGELU = 2,  /* bf16 out, x*sigmoid(1.702x) (CLIPMLP 'quick_gelu')                              */
  VT_EPI_BF16_RELU = 3,   /* bf16 out, ReLU             (region_extractor MLP / LocationEncoder)            */
  VT_EPI_F32_RESID = 4,   /* fp32 C += A.W^T + bias      (residual stream accumulate)                        */
  VT_EPI_F32 = 5,         /* fp32 out                    (logits)                                            */
  VT_EPI_SWIGLU_BF16 = 6  /* W rows = [gate16|up16|gate16|up16...]; bf16 out [M][N/2] = silu(gate)*up          */
};
/* tile configurations of the MFMA GEMM (VT_GEMM_CFG_AUTO lets the library choose) */
enum {
  VT_GEMM_CFG_AUTO = 0,
  VT_GEMM_CFG_SKINNY = 1,   /* M <= 16 weight-streaming kernel (per-wave LDS-DMA ring -> MFMA, split-K inside the block) */
  VT_GEMM_CFG_128x128 = 2,
  VT_GEMM_CFG_256x128 = 3,
  VT_GEMM_CFG_256x256 = 4,
  VT_GEMM_CFG_64x128 = 5,
  VT_GEMM_CFG_256x256_P8 = 6, /* 256x256 tile, 8 waves, 8-phase pipelined main loop */
  /* 7: reserved (a 4-wave 128x128-register-tile experiment, slower than the 8-phase kernel; removed) */
  VT_GEMM_CFG_256x256_RP = 8, /* 256x256 tile, 8 waves, register-pipelined 32x32x16 main loop, one barrier per K tile */
  VT_GEMM_CFG_SKINNY_REG = 9, /* M <= 16: weight rows loaded straight into MFMA operand registers; fallback when K % 64 != 0 */
  VT_GEMM_CFG_256x256_P4 = 10, /* the 8-phase kernel with its phases merged pairwise: 32 MFMAs per section, half the barriers */
  /* 11: reserved (the 4-phase schedule on v_mfma_f32_32x32x16_bf16: correct, 25 % slower on every decoder shape; removed) */
  /* 12: reserved (2-phase schedule, one 64-MFMA section per K step: correct, 8 % slower than the 4-phase one; removed) */
  VT_GEMM_CFG_256x256_W4 = 13, /* 256x256 tile, 4 waves x (128x128), one wave per SIMD, K step placed by hand (plain-store epilogues) */
  VT_GEMM_CFG_320x256_W4 = 14, /* the same kernel on 320-row tiles (160x128 per wave): rows that fill whole rounds only this way */
  VT_GEMM_CFG_160x128_W4 = 15, /* four waves of 80x64 on a 160x128 tile, four-deep LDS ring: N = 1024 projections (K % 256 == 0) */
  VT_GEMM_CFG_224x256_W4 = 16, /* the four-wave kernel on 224-row tiles (112x128 per wave): row counts a little over a multiple of 224 (1088, 4616) */
};
enum { VT_DTYPE_BF16 = 0, VT_DTYPE_OP16 = 0 /* the 16-bit operand format of the library */, VT_DTYPE_F32 = 1 };
enum { VT_ACT_GELU = 0, VT_ACT_QUICK_GELU = 1 };

#define VT_PAGE_TOKENS 64 /* tokens per KV-cache page == keys per attention tile */

#define VT_ABI_VERSION 113
int vt_version(void); /* == VT_ABI_VERSION of the header the library was built from */
/* operand format of THIS library (see Conventions): every uint16_t tensor argument carries these bits */
enum { VT_OPERAND_BF16 = 0, VT_OPERAND_FP16 = 1 };
int vt_operand_format(void);
/* copies the calling thread's last error message (NUL terminated) into buf; returns its length */
int vt_last_error(char* buf, size_t buf_len);

/* ------------------------------------------------------------------------------------------------------------
 * Primitive operators (exported for unit parity tests and for composing other models)
 * ---------------------------------------------------------------------------------------------------------- */

/* nn.Linear: C = epi(A[M,K] . W[N,K]^T + bias[N]).  bias may be NULL. C is bf16 or fp32 per `epi`.
 * `row_scale` (optional, fp32 [M], device): the accumulator of row m is multiplied by row_scale[m] before bias and activation
 * -- a per-row scalar commutes with the contraction, so a GEMM on x .* w followed by this factor is Linear(RMSNorm(x)) with the
 * normalisation folded in (what vt_llama_forward does for rows > 64). Rows with a scale run on the MFMA tile kernels whatever M
 * is. NULL = no scaling.
 * Replaces every torch.nn.Linear on the path (transformers-4.31 CLIPAttention/CLIPMLP/LlamaAttention/LlamaMLP,
 * reference call sites modeling_video.py:69,71,81; llava_llama.py:49,91-102; multimodal_projector/builder.py:33-51). */
int vt_gemm_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, void* C, int ldc, const float* bias, int M,
                 int N, int K, int epi, int cfg, const float* row_scale, void* stream);

/* Which tile configuration vt_gemm_bf16(.., cfg = VT_GEMM_CFG_AUTO, row_scale = NULL) runs for this shape and epilogue (host logic
 * only, no launch): *cfg = the VT_GEMM_CFG_* of the first launch, *rows_first = 0 when that launch covers all M rows, else the
 * number of leading rows it covers (the remaining rows are planned again, as a GEMM of their own). Introspection for tests and
 * tools; no reference counterpart (torch.nn.Linear hides cuBLAS's heuristics the same way). */
int vt_gemm_plan_query(int M, int N, int K, int epi, int* cfg, int* rows_first);
/* the same with the column split of round 5: *cols_first > 0 means columns [0, *cols_first) run on *cfg (whole rounds of 256-row tiles) and the
 * remaining columns are planned again (a few row blocks x many column tiles that spill just over whole rounds: 768 x 22016) */
int vt_gemm_plan_query2(int M, int N, int K, int epi, int* cfg, int* rows_first, int* cols_first);

/* y_bf16[rows][D] = LayerNorm(x_f32) * gamma + beta. If temb != NULL first x[row] += temb[(row / tokens_per_frame) % T]
 * (written back): the video tower's temporal_embedding add (reference modeling_video.py:110-114) fused with
 * temporal_layer_norm1 (:117). */
int vt_layernorm(float* x, const float* temb, int T, int tokens_per_frame, const float* gamma, const float* beta,
                 uint16_t* y, int rows, int D, float eps, void* stream);

/* y_bf16[r] = w * x[idx ? idx[r] : r] * rsqrt(mean(x^2) + eps) : transformers-4.31 LlamaRMSNorm. */
int vt_rmsnorm(const float* x, const int* idx, const float* w, uint16_t* y, int rows, int D, float eps, void* stream);

/* ---- precise level 3: the rounding remainder of a GEMM's A operand on the MX-FP4 pipe (no reference counterpart: the reference computes
 * every Linear in ONE 16-bit format, vitron/model/builder.py:47; north_star's 1e-3 against its fp32 CPU path is what these serve) ----------
 * A Linear of level 3 is  C = epi(A.W^T + (A4 2^ea).(W4 2^ew)^T + bias): A = op16(v) against the 16-bit weights on the 16-bit MFMA, plus
 * A4 = the MX-FP4 image (OCP e2m1 elements, power-of-two block scales) of lo = v - f32(A) against the weights' MX-FP4 image on
 * v_mfma_scale_f32_16x16x128_f8f6f4, in ONE launch and one fp32 accumulator.
 *   W4   uint8 [N][K/2]: element k of row n in byte k/2 (even k in bits 3:0);  wexp uint8 [N]: one biased (e8m0) exponent per row
 *   A4   uint8 [M][K/2]: same element order;  aexp uint8 [vt_mx4_aexp_bytes(M, K)]: one biased exponent per row and 32 consecutive k, at
 *        aexp[((m / 64) * (K / 32) + kb) * 64 + (m % 16) * 4 + (m % 64) / 16]  (the order the GEMM's lanes fetch them; rows padded to 256)
 * Quantisation: scale 2^e with the smallest e for which the block's largest magnitude is <= 6, elements to the nearest e2m1 value
 * {0, .5, 1, 1.5, 2, 3, 4, 6}, ties to the even mantissa (oracle/vitron_oracle.py mx4_quant restates it; tests compare bit for bit). */
size_t vt_mx4_aexp_bytes(int M, int K);
/* weights (K % 8 == 0) */
int vt_mx4_quant_weights(const uint16_t* W, int ldw, int N, int K, uint8_t* W4, uint8_t* wexp, void* stream);
/* a 16-bit remainder lo [M][K] (the second output of the precise level 2 operators), K % 32 == 0 */
int vt_mx4_quant_lo(const uint16_t* lo, int ld, int M, int K, uint8_t* A4, uint8_t* aexp, void* stream);
/* vt_rmsnorm with the level 3 operand out: y = op16(v), (A4, aexp) = MX-FP4 image of v - f32(y); D % 512 == 0 */
int vt_rmsnorm_mx(const float* x, const int* idx, const float* w, uint16_t* y, uint8_t* A4, uint8_t* aexp, int rows, int D, float eps,
                  void* stream);
/* the GEMM: 256 x 256 tiles of the four-wave kernel, K % 128 == 0, K >= 256, N % 4 == 0; epi = VT_EPI_BF16 / _GELU / _QGELU / F32_RESID /
 * F32 / SWIGLU_BF16 */
int vt_gemm_mx(const uint16_t* A, int lda, const uint8_t* A4, const uint8_t* aexp, const uint16_t* W, int ldw, const uint8_t* W4,
               const uint8_t* wexp, void* C, int ldc, const float* bias, int M, int N, int K, int epi, void* stream);

/* gate/up of level 3: C [M][N/2] = op16(v), v = silu(gate) * up over the 16-interleaved columns, and (out4 [M][N/4], oexp) = the MX-FP4 image of
 * v - f32(C) -- down_proj's second operand, written by the same launch. N % 128 == 0. */
int vt_gemm_mx_swiglu(const uint16_t* A, int lda, const uint8_t* A4, const uint8_t* aexp, const uint16_t* W, int ldw, const uint8_t* W4,
                      const uint8_t* wexp, uint16_t* C, int ldc, uint8_t* out4, uint8_t* oexp, int M, int N, int K, void* stream);
/* the towers' fc1 in their precise level 1: C [M][N] = op16(v), v = gelu(acc + bias) (quick = 0: erf form, 1: x sigmoid(1.702 x)), and
 * (out4 [M][N/2], oexp) = the MX-FP4 image of v - f32(C) -- fc2's second operand. N % 64 == 0. */
int vt_gemm_mx_gelu(const uint16_t* A, int lda, const uint8_t* A4, const uint8_t* aexp, const uint16_t* W, int ldw, const uint8_t* W4,
                    const uint8_t* wexp, const float* bias, uint16_t* C, int ldc, uint8_t* out4, uint8_t* oexp, int M, int N, int K, int quick,
                    void* stream);
/* vt_layernorm (without the temporal-embedding add) with the level 3 operand out, as vt_rmsnorm_mx; D % 256 == 0 */
int vt_layernorm_mx(const float* x, const float* gamma, const float* beta, uint16_t* y, uint8_t* A4, uint8_t* aexp, int rows, int D, float eps,
                    void* stream);
/* x += A.W^T + A4.W4^T (o_proj, down_proj of level 3). A grid that spills a fraction of a round over whole rounds of the chip's 256
 * multiprocessors runs its trailing row blocks as K ranges on the idle ones (fp32 partial slabs in `partials`, then an ordered reduce);
 * partials may be NULL (one plain launch). */
int vt_gemm_mx_resid(const uint16_t* A, int lda, const uint8_t* A4, const uint8_t* aexp, const uint16_t* W, int ldw, const uint8_t* W4,
                     const uint8_t* wexp, float* C, int ldc, int M, int N, int K, float* partials, size_t partial_bytes, void* stream);
/* vt_flash_attn (causal, head_dim 128, dense output: ldo == heads * 128) that also writes (O4 [rows][heads * 64], oexp) = the MX-FP4 image of
 * the output's rounding remainder -- o_proj's second operand in level 3. */
int vt_flash_attn_mx(const uint16_t* Q, int ldq, const uint16_t* k_tiles, const uint16_t* vt_tiles, const int* tile_table,
                     const int* seq_desc, int nseq, int max_q_len, uint16_t* O, int ldo, uint8_t* O4, uint8_t* oexp, int heads, float scale,
                     void* stream);

/* Attention over 64-key tiles. seq_desc: device int32 [nseq][4] = {q_row0, q_len, kv_len, table_off};
 * tile_table: device int32, tile_table[table_off + t] = index of the sequence's t-th K / V^T tile.
 * K tile = [64 keys][HD] bf16, V^T tile = [HD][64 keys] holding IEEE fp16 bits (the bf16 values of the projection converted
 * exactly, saturating at +-65504: the P.V product of the prefill kernel runs on the f16 MFMA with P at 11 mantissa bits; only
 * vt_kv_tiles / vt_attn_decode_fused / the fused QKV epilogue write these tiles); tile i of head h starts at (i*heads + h)*64*HD
 * elements.
 * q_len > 1 anywhere -> MFMA flash kernel (causal: query i sees keys <= kv_len - q_len + i); all q_len == 1 is
 * routed to the single-query kernel by vt_llama_forward. Replaces CLIPAttention (non-causal, modeling_video.py:136-146)
 * and LlamaAttention's softmax(QK^T/sqrt(d) + mask)V (reference restatement: llama_flash_attn_monkey_patch.py:30-66). */
int vt_flash_attn(const uint16_t* Q, int ldq, const uint16_t* k_tiles, const uint16_t* vt_tiles, const int* tile_table,
                  const int* seq_desc, int nseq, int max_q_len, uint16_t* O, int ldo, int heads, int head_dim,
                  int causal, float scale, void* stream);
/* Which prefill kernel vt_flash_attn (and the composite operators) run for head_dim 128 -- a process-wide switch for A/B measurements
 * and for the tests that hold the kernels to each other; results are the same contract either way. 0 = automatic (the one-wave-per-SIMD
 * kernel on 256-row blocks once heads x sequences x blocks fills the 256 CUs, else the two-waves-per-SIMD kernel on 128-row blocks),
 * 1 = always the two-waves-per-SIMD kernel, 2 = always the one-wave-per-SIMD kernel, 3 = that kernel with its pipeline stages run one
 * after the other instead of the hand-placed schedule (its reference), 4 = the one-wave-per-SIMD kernel in its persistent form (one
 * workgroup per CU walking the block list, K / V^T stream running on across block seams; `kernel = 4 | (n << 8)` caps it at n
 * workgroups -- a test hook that forces many blocks per workgroup). The persistent form keeps its block tickets in device globals:
 * one launch of it at a time per library (launches on one stream are ordered). 5 = kernel 2 dispatched in the grid's natural block order
 * instead of the planned one (next entry). No reference counterpart. */
int vt_flash_attn_select(int kernel);
/* Host logic only (no device work): the order in which a CAUSAL launch of the one-wave-per-SIMD kernel hands its (head, 256-row block,
 * sequence) blocks to the hardware dispatcher. order[i] = head + heads * (y + q_blocks * sequence) for workgroup i, y = 0 the last (heaviest)
 * block of a sequence. The dispatcher gives workgroup i to the first CU that frees up, so the order IS the schedule: the natural order
 * (heaviest first) is LPT; for a single long sequence a bin packing at a target makespan + "list by planned start time" ends 2.9 % above
 * the mean CU load instead of 11.8 % (S = 5120, 32 heads, 256 CUs). Planned per XCD (entry i runs on XCD i % 8 and only names heads = i mod
 * 8). Returns the number of entries written (heads * q_blocks * nseq), or 0 when the natural order is kept (one round, very many blocks, or
 * a plan that does not beat it by 2 % under the cost model). No reference counterpart. */
int vt_flash_attn_block_order(int heads, int q_blocks, int nseq, int multiprocessors, int* order, int capacity);
/* Residual GEMM C[M,N] (fp32) += A[M,K] W[N,K]^T + bias with an optional split-K workspace: when the 256x256 tile grid would
 * cover at most half of the chip (M ~ 1000 rows at N = 4096, the ViT's N = 1024 projections) the K loop is split over up to 8
 * workgroups per tile, fp32 partial products go to `partials` and a second kernel adds them in split order (deterministic).
 * ksplit: 0 = decide (may not split), >= 2 = force; partial_bytes >= ksplit*M*N*4 when it splits. */
int vt_gemm_bf16_resid_splitk(const uint16_t* A, int lda, const uint16_t* W, int ldw, float* C, int ldc, const float* bias,
                              int M, int N, int K, int ksplit, float* partials, size_t partial_bytes, void* stream);

/* One decode step's attention in one launch (every sequence has q_len == 1; kv_len counts the new token): rotary
 * embedding of the new q/k rows (rope tables may be NULL), append of the new k row / v column to the paged tiles (a tile
 * that starts with the new token is zero-filled around it), single-query attention over the cache. qkv is not modified.
 * Equivalent to vt_kv_tiles + vt_attn_decode; replaces LlamaAttention's cached single-token step
 * (transformers 4.31 LlamaAttention.forward as called from vitron/model/language_model/llava_llama.py:91-102). */
int vt_attn_decode_fused(const uint16_t* qkv, int ldqkv, int q_col0, int k_col0, int v_col0, uint16_t* k_tiles,
                         uint16_t* vt_tiles, const int* tile_table, const int* seq_desc, int nseq, uint16_t* O, int ldo,
                         int heads, int head_dim, float scale, const float* rope_cos, const float* rope_sin,
                         const int* positions, void* stream);

/* single-query (q_len == 1) attention, split over the KV tiles; scratch >= vt_attn_decode_scratch_bytes(...) */
size_t vt_attn_decode_scratch_bytes(int nseq, int heads, int head_dim, int max_kv_len);
int vt_attn_decode(const uint16_t* Q, int ldq, const uint16_t* k_tiles, const uint16_t* vt_tiles, const int* tile_table,
                   const int* seq_desc, int nseq, uint16_t* O, int ldo, int heads, int head_dim, float scale,
                   int max_kv_len, void* scratch, size_t scratch_bytes, void* stream);
/* fused-QKV rows -> K tiles / V^T tiles for the NEW tokens of each sequence; when rope_cos != NULL applies the
 * half-split rotary embedding (tables [rope_len][HD/2] fp32, row = positions[row]) to k and, in place, to q.
 * Replaces apply_rotary_pos_emb + the torch.cat KV-cache append of transformers-4.31 LlamaAttention. */
int vt_kv_tiles(uint16_t* qkv, int ldqkv, int q_col0, int k_col0, int v_col0, uint16_t* k_tiles, uint16_t* vt_tiles,
                const int* tile_table, const int* seq_desc, int nseq, int max_new_tiles, int heads, int head_dim,
                const float* rope_cos, const float* rope_sin, const int* positions, void* stream);
/* temporal attention of the video tower: qkv bf16 [B*T*N][3*heads*64] (row (b*T+t)*N+n, q pre-scaled) ->
 * out bf16 [B*T*N][heads*64]; attends over the T frames at each (b, n). Reference modeling_video.py:105-127. */
int vt_attn_temporal(const uint16_t* qkv, uint16_t* out, int B, int T, int N, int heads, void* stream);

/* CLIPVisionEmbeddings patch gather: pixels ([F][3][H][W] or, video_layout=1, [B][3][T][H][W]; bf16 or fp32) ->
 * patches bf16 [B*T*(H/P)*(W/P)][k_pad], K order (c,py,px), zero padded. */
int vt_im2col(const void* pixels, int pix_dtype, uint16_t* patches, int B, int T, int H, int W, int P, int k_pad,
              int video_layout, void* stream);

/* Pre-processing of decoded frames on the device, fused: short-side resize to S (bicubic = torchvision tensor Resize,
 * reference image/processing_image.py:15-25; bilinear = pytorchvideo ShortSideScale, video/processing_video.py:45-53),
 * centre crop SxS, *1/255 for uint8 sources, (x - mean) / std, optional horizontal flip.
 * src: F frames, [F][H][W][3] (hwc=1) or [F][3][H][W]; uint8 (src_u8=1) or fp32. mean/std: HOST float[3].
 * dst element (c, f, y, x) at c*dst_stride_c + f*dst_stride_f + y*S + x  (images: sc=S*S, sf=3*S*S; clips [3][T][S][S]:
 * sc=T*S*S, sf=S*S); bf16 or fp32. */
int vt_preprocess(const void* src, int src_u8, int hwc, int F, int H, int W, int bicubic, int S, const float* mean,
                  const float* std, int flip, void* dst, int dst_dtype, long dst_stride_c, long dst_stride_f, void* stream);

/* embed_tokens gather + visual / region splice (reference llava_arch.py:306-398, 479-558).
 * plan: device int32 [rows][2] = {kind, index}; kind 0 = token id, 1 = row of vis, 2 = row of reg, 3 = zero row.
 * tok_table has `vocab` rows, vis `vis_rows`, reg `reg_rows` (0 with a NULL table): an index outside its table gives a
 * zero row instead of an out-of-bounds read (the host mirror raises before it gets that far). */
int vt_embed_splice(const uint16_t* tok_table, int vocab, const uint16_t* vis, int vis_rows, const uint16_t* reg, int reg_rows,
                    const int* plan, int rows, int H, uint16_t* out, void* stream);

/* greedy next token: first index of the row maximum (GenerationMixin greedy search). */
int vt_argmax(const float* logits, int rows, int V, int ldl, int* out_ids, void* stream);

/* Decode-loop feedback, one launch per generated token (the body of GenerationMixin's sampling loop that
 * model.generate runs between two forward passes, reference app.py:562-571 / inference_image.py:60-70 via
 * transformers; here without the host round trip). For every sequence i < nseq:
 *   t = finished[i] ? pad_id : next_ids[i];  finished[i] |= (t in eos_ids[0..n_eos));
 *   tokens_out[i] = t;  tokens_out[nseq + i] = finished[i];          (int32 [2][nseq], for the asynchronous read-back)
 *   x[i][0..H) = tok_table[t][0..H)                                  (input row of the next vt_llama_forward)
 *   seq_desc[i].kv_len += 1;  positions[i] += 1                      (device-side step metadata of that pass)
 * so the next decoder pass can be enqueued before the host has seen the token. */
int vt_decode_feed(const uint16_t* tok_table, int H, int vocab, const int* next_ids, int* finished, const int* eos_ids, int n_eos,
                   int pad_id, int* tokens_out, uint16_t* x, int* seq_desc, int* positions, int nseq, void* stream);
/* Mean cross entropy over the rows whose label is not `ignore_index`: loss[0] = mean_r(logsumexp(logits[r]) - logits[r][labels[r]]).
 * The caller shifts (row r scores the token at position r + 1). row_nll: fp32 scratch [rows] (per-row losses, -1 = skipped).
 * Replaces the CrossEntropyLoss of LlamaForCausalLM.forward when `labels` are passed (reference llava_llama.py:91-102). */
int vt_cross_entropy(const float* logits, int rows, int V, int ldl, const int* labels, int ignore_index, float* row_nll,
                     float* loss, void* stream);

/* one sampled token per row, the warper chain of GenerationMixin.sample in transformers' order: logits / temperature,
 * TopKLogitsWarper (top_k > 0: everything below the k-th largest value is removed, ties stay; 0 = off), softmax,
 * TopPLogitsWarper keep-set (top_p >= 1 keeps all), inverse-CDF draw with a counter-based uniform of (seed, step, row).
 * kept_count (optional) = size of the final keep-set per row.
 * Replaces GenerationMixin.sample's temperature / top-k / top-p / multinomial step (reference app.py:562-571, do_sample=True). */
int vt_sample_top_p(const float* logits, int rows, int V, int ldl, float temperature, int top_k, float top_p, uint64_t seed,
                    uint64_t step, int* out_ids, int* kept_count, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * mm_projector: Linear(Din,Dh) -> GELU(erf) -> Linear(Dh,Dout)   ('mlp2x_gelu';  w2 == NULL -> 'linear')
 * reference vitron/model/multimodal_projector/builder.py:33-51, called at llava_arch.py:175,186
 * ---------------------------------------------------------------------------------------------------------- */
size_t vt_projector_workspace_bytes(int M, int Dh);
/* precise level 2 (mlp2x_gelu only): features in and embeddings out as operand pairs (x + x_lo, out + out_lo: hi = op(v), lo = op(v - hi)),
 * both Linear layers as A_hi.W^T + A_lo.W^T in fp32, GELU on the fp32 value. reference multimodal_projector/builder.py:40-46 */
size_t vt_projector_precise_workspace_bytes(int M, int Dh, int Dout);
int vt_projector_forward_precise(const uint16_t* x, const uint16_t* x_lo, int M, int Din, const uint16_t* w1, const float* b1, int Dh,
                                 const uint16_t* w2, const float* b2, int Dout, uint16_t* out, uint16_t* out_lo, void* workspace,
                                 size_t workspace_bytes, void* stream);
int vt_projector_forward(const uint16_t* x, int M, int Din, const uint16_t* w1, const float* b1, int Dh,
                         const uint16_t* w2, const float* b2, int Dout, uint16_t* out, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * region_extractor (reference vitron/model/region_extractor/layer.py:58-130), one box per image.
 *   feats   bf16 [B][G*G][D]      patch features of the image tower (pre-projector)
 *   slices  int32 [B][4]          {row_start,row_stop,col_start,col_stop} = Python slice(int(x1),int(x2)) /
 *                                  slice(int(y1),int(y2)) resolved against image_size on the host (x -> rows, :83)
 *   coords  fp32 [B][4]           raw box {x1,y1,x2,y2} (LocationEncoder input, :126; fp32 so that canvas coordinates above 256,
 *                                  which bf16 cannot hold, reach the K = 4 layer unrounded -- the reference casts them to the
 *                                  model dtype, where fp16 holds integers to 2048)
 *   mlp_w[2]/mlp_b[2]             first two layers of region_linear (D->H, H->H, ReLU after each; layer.py:17-20)
 *   loc_w0 / loc_b0               LocationEncoder layer 0, bf16 [H/2][8] (zero padded from [H/2][4]) / fp32 [H/2]
 *   final_w / final_b             [H][H + H/2] = [ region_linear.layers.2.weight | loc_encoder.2.weight ] and the SUM of their
 *                                  biases: MLP output + location embedding (layer.py:129) is one contraction over the
 *                                  concatenated K, accumulated in fp32 and rounded to bf16 once
 *   out     bf16 [B][H]; cell_mask int32 [B][G*G] and cell_count int32 [B] are optional outputs (may be NULL)
 *   B <= 16 per call, G <= 64. Four launches: pool (+ mask, count, LocationEncoder layer 0) and three weight-streaming GEMMs.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct vt_region_weights {
  int in_dim, out_dim;
  const uint16_t* mlp_w[2];
  const float* mlp_b[2];
  const uint16_t* loc_w0;
  const float* loc_b0;
  const uint16_t* final_w;
  const float* final_b;
} vt_region_weights;
size_t vt_region_workspace_bytes(int B, int in_dim, int out_dim);
int vt_region_forward(const vt_region_weights* w, const uint16_t* feats, const int* slices, const float* coords,
                      int B, int G, int image_size, uint16_t* out, int* cell_mask, int* cell_count, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * LanguageBind ViT tower (CLIP ViT + optional per-layer temporal attention)
 * reference CLIPVisionTransformer.forward / CLIPEncoderLayer.forward
 *   vitron/model/multimodal_encoder/languagebind/video/modeling_video.py:65-158,596-675 (image twin: image/modeling_image.py)
 * and the tower wrappers' feature_select (languagebind/__init__.py:96-121,182-204): the library runs `num_layers`
 * encoder layers (= n_hidden + 1 + select_layer, 23 of 24 for select_layer = -2) and returns the patch tokens
 * (CLS dropped) of that hidden state.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct vt_vit_layer {
  /* temporal block -- all NULL unless add_time_attn */
  const float *t_ln_g, *t_ln_b;
  const float* t_embed;   /* [num_frames][D]; NULL when num_frames == 1 (reference skips the add, :110) */
  const uint16_t* t_wqkv; /* [3D][D], q rows pre-scaled by head_dim^-0.5 */
  const float* t_bqkv;    /* [3D], q part pre-scaled */
  const uint16_t* t_wo;
  const float* t_bo;
  /* spatial block */
  const float *ln1_g, *ln1_b;
  const uint16_t* wqkv;
  const float* bqkv;
  const uint16_t* wo;
  const float* bo;
  const float *ln2_g, *ln2_b;
  const uint16_t* w1; /* fc1 [I][D] */
  const float* b1;
  const uint16_t* w2; /* fc2 [D][I] */
  const float* b2;
  /* temporal MLP -- the IMAGE tower's add_time_attn variant only (reference image/modeling_image.py:83-84,129-134: after the temporal
     attention, x += temporal_mlp(temporal_layer_norm2(x))); all NULL for the video tower (video/modeling_video.py has no such block) */
  const float *t_ln2_g, *t_ln2_b;
  const uint16_t* t_w1;
  const float* t_b1;
  const uint16_t* t_w2;
  const float* t_b2;
  /* precise level 1 only, else NULL: MX-FP4 images of fc1 / fc2 (vt_mx4_quant_weights) and their per-row exponents -- with them the MLP runs
     as two launches of vt_gemm_mx's kernel (hidden % 256 == 0, intermediate % 128 == 0); without them level 1 falls back to 16-bit operand pairs */
  const uint8_t* w14; const uint8_t* w1_e;
  const uint8_t* w24; const uint8_t* w2_e;
} vt_vit_layer;

typedef struct vt_vit_model {
  int image_size, patch, hidden, heads, intermediate;
  int num_layers;    /* encoder layers to run */
  int num_frames;    /* T of the temporal attention (config.num_frames) */
  int add_time_attn; /* 1 = video tower */
  int act;           /* VT_ACT_* */
  float ln_eps;
  int k_pad;               /* padded K of the patch GEMM (multiple of 64, >= 3*patch*patch) */
  const uint16_t* w_patch; /* [D][k_pad] */
  const float* cls;        /* [D] */
  const float* pos;        /* [G*G+1][D] */
  const float *pre_ln_g, *pre_ln_b;
  const vt_vit_layer* layers; /* host array [num_layers] */
  int precise;             /* 2: precise level 2 -- EVERY GEMM A operand of the tower travels as an operand pair (hi + lo), every product as
                              two launches accumulating in fp32: the norm outputs, q and k through the attention scores (the decoder's
                              precise kernels at head_dim 64; v goes from fp32 straight into the V^T tiles' fp16), the attention outputs,
                              the temporal attention in fp32, the activation outputs; with out_feats_lo the selected patch tokens leave
                              as a pair too. The hidden state is then within ~1e-5 of fp32 in both operand builds.
                              1: the MLPs' operands (layer_norm2 -> fc1, activation -> fc2: what carries the tower's distance from fp32)
                              with their rounding remainders -- as MX-FP4 images in the same launch when the layers carry w14 / w24
                              (else as 16-bit operand pairs, two launches) -- and out_feats as a pair; the attention paths standard:
                              the towers' share of the decoder's precise level 3.
                              0 (default): standard. */
  uint16_t* out_feats_lo;  /* optional DEVICE buffer [B*T*G*G][D]: the low half of out_feats (precise >= 1 only) */
} vt_vit_model;

size_t vt_vit_workspace_bytes(const vt_vit_model* m, int B, int T);
/* pixels: image tower [B][3][H][W] (T = 1, video_layout = 0); video tower [B][3][T][H][W] (video_layout = 1).
 * out_feats  bf16 [B*T*G*G][D]  patch tokens of the selected hidden state (required)
 * out_hidden fp32 [B*T*(G*G+1)][D] the full selected hidden state incl. CLS (optional, NULL to skip) */
int vt_vit_forward(const vt_vit_model* m, const void* pixels, int pix_dtype, int B, int T, int video_layout,
                   uint16_t* out_feats, float* out_hidden, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * LLaMA decoder (Vicuna-7B shape class): prefill and decode through one entry point, paged KV cache.
 * reference LlavaLlamaForCausalLM.forward -> transformers-4.31 LlamaForCausalLM.forward
 *   vitron/model/language_model/llava_llama.py:57-102 ; decode-step plumbing llava_arch.py:196-205
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct vt_llama_layer {
  const float* rms1;     /* input_layernorm.weight [H] */
  const uint16_t* wqkv;  /* [3H][H] rows = q | k | v */
  const uint16_t* wo;    /* [H][H] */
  const float* rms2;     /* post_attention_layernorm.weight */
  const uint16_t* wgu;   /* [2I][H], rows interleaved in blocks of 16: gate | up | gate | up ... */
  const uint16_t* wdown; /* [H][I] */
  /* precise level 3 (precise_qk = 3) only, else NULL: the MX-FP4 images of the four weight matrices (vt_mx4_quant_weights on the packed
   * matrices above: same row order) and their per-row exponents */
  const uint8_t* wqkv4;  const uint8_t* wqkv_e;
  const uint8_t* wo4;    const uint8_t* wo_e;
  const uint8_t* wgu4;   const uint8_t* wgu_e;
  const uint8_t* wdown4; const uint8_t* wdown_e;
} vt_llama_layer;

typedef struct vt_llama_model {
  int hidden, heads, head_dim, intermediate, num_layers, vocab;
  float rms_eps;
  const float* final_norm;    /* [H] */
  const uint16_t* lm_head;    /* [V][H] */
  const float* rope_cos;      /* [rope_len][head_dim/2] */
  const float* rope_sin;
  int rope_len;
  const vt_llama_layer* layers; /* host array [num_layers] */
  int prefill_norm_fold;        /* 1: prefills (rows > 64) fold RMSNorm into the MFMA tile GEMMs (residual epilogues emit bf16(x .* w)
                                   and partial sums of x^2, the consumer GEMMs scale their rows); same accuracy, measured no faster:
                                   default 0. Decode steps (<= 16 rows) always fold. */
  int qkv_fuse;                 /* 1: prefills write rotated q / K pages / V^T pages from the QKV projection's epilogue instead of the
                                   separate vt_kv_tiles pass. Bit-identical results; measured 20 us per launch SLOWER at S = 5120 (the
                                   epilogue of a one-workgroup-per-CU kernel overlaps with nothing: 455 vs 386 + 48.5 us): default 0. */
  int precise_qk;               /* 3: precise level 3 (head_dim 128, hidden % 512 == 0, intermediate % 128 == 0; the layers' *4 / *_e images set):
                                   PREFILLS run every decoder Linear as ONE launch that adds, to the 16-bit product, the product of the MX-FP4
                                   image of the A operand's rounding remainder with the weights' MX-FP4 image on the 4x-rate MX pipe (vt_gemm_mx;
                                   RMSNorm, the attention kernels and the SwiGLU epilogue emit the remainder's image beside the 16-bit operand);
                                   the lm_head's operand is a 16-bit pair as in level 2; q / k / v / P / V^T are stored as in the standard mode.
                                   The mode north_star's 1e-3 is met in at ~1.25x of the standard step. A request the shapes
                                   cannot serve is an ERROR, never a silent standard-mode pass.
                                   2: precise level 2 -- level 1 below plus EVERY other GEMM A operand of a prefill as a pair (v projection,
                                   attention output -> o_proj, post-attention norm -> gate/up with the SwiGLU as its own fp32 -> pair pass,
                                   SwiGLU output -> down_proj, final norm -> lm_head): each product runs
                                   as two launches accumulating in fp32. A verification mode: ~2x the GEMM work.
                                   1 (head_dim 128): PREFILLS carry everything that reaches the softmax's argument as operand PAIRS
                                   (hi + lo, 2 x 16 bit): the input-norm output feeds the q / k projection as A_hi.W^T + A_lo.W^T into fp32,
                                   the rotary embedding runs in fp32, and the attention scores are K_hi.(Q_hi + Q_lo)^T + K_lo.Q_hi^T -- so the
                                   scores see ~2^-20 of operand rounding instead of 2^-12 (fp16) / 2^-9 (bf16). Removes the five storage points
                                   the softmax amplifies: fp16 full-depth logits 1.3e-3 -> below 1e-3 of the reference's fp32.
                                   Costs two extra GEMM launches, +2 MFMAs per score k-step and the workspace for the pairs; the K pages hold
                                   K_hi (decode steps and later passes are unchanged). Default 0. */
  const uint16_t* embeds_lo;    /* optional DEVICE buffer [rows][H] in the operand format: the LOW half of the input embeddings when the caller
                                   carries them as a pair (precise level 2: the projector's output is not rounded to 16 bits on its way into the
                                   residual stream); added to x_embeds in fp32. NULL (default): x_embeds alone. */
  float* hidden_trace;          /* optional DEVICE buffer fp32 [num_layers][rows][H]: when non-NULL every layer's INPUT residual stream is
                                   copied there (entry 0 = the input embeddings) -- what `output_hidden_states=True` of the reference's
                                   LlamaModel collects before each decoder layer (llava_llama.py:69, transformers 4.31 LlamaModel.forward).
                                   NULL (default): nothing is copied. */
} vt_llama_model;

/* KV pool: k  [num_layers][num_pages][heads][64][head_dim]   (K rows, rotary applied)
 *          vt [num_layers][num_pages][heads][head_dim][64]   (V transposed inside the page) */
typedef struct vt_kv_cache {
  uint16_t* k;
  uint16_t* vt;
  int num_pages;
} vt_kv_cache;

size_t vt_llama_workspace_bytes(const vt_llama_model* m, int rows, int n_logit_rows, int nseq, int max_kv_len);
/* x_embeds    bf16 [rows][H]   packed input embeddings of all sequences (no padding rows)
 * positions   int32 [rows]     rotary position of every row
 * seq_desc    int32 [nseq][4]  {q_row0, q_len, kv_len, table_off} ; kv_len = past + q_len
 * tile_table  int32            page ids, tile_table[table_off + t] = page of positions [64t, 64t+64)
 * logit_rows  int32 [n_logit_rows] rows whose logits are wanted (generate: the last row of every sequence)
 * logits      fp32 [n_logit_rows][V]
 * The new tokens' K/V are appended to the cache pages; attention is causal over past + new tokens. */
int vt_llama_forward(const vt_llama_model* m, const vt_kv_cache* kv, const uint16_t* x_embeds, int rows,
                     const int* positions, const int* seq_desc, int nseq, int max_q_len, int max_new_tiles, int max_kv_len,
                     const int* tile_table, const int* logit_rows, int n_logit_rows, float* logits,
                     float* out_hidden, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Per-kernel-class timing (the reference has no tracing on this path). Between vt_profile_begin and
 * vt_profile_end every launch of the instrumented kernel classes is bracketed by hipEvents recorded on the SAME
 * stream as the kernel, so the durations are device-side launch durations inside the caller's timed region.
 * vt_profile_end synchronises on the recorded events and fills, per class: launches, total milliseconds, total
 * algorithmic work (FLOP for MFMA kernels, bytes for the weight-streaming kernel). Arrays hold VT_PROF_CLASSES items.
 * ---------------------------------------------------------------------------------------------------------- */
enum { VT_PROF_GEMM_TILE = 0, VT_PROF_FLASH_ATTN = 1, VT_PROF_GEMM_SKINNY = 2, VT_PROF_ATTN_DECODE = 3, VT_PROF_CLASSES = 4 };
int vt_profile_begin(void);
int vt_profile_end(int* launches, double* total_ms, double* total_work);
/* What the matrix pipe sustains by itself on THIS device (measurement aid for the roofline, no reference counterpart): one wave per SIMD
 * on every CU, operands in registers, nothing but v_mfma_f32_16x16x32 (bf16 or f16: the library's operand format) on a 128 x 128
 * accumulator tile in the loop -- the four-wave GEMM's wave tile without its memory side. a, b: at least 64 Ki 16-byte fragments each
 * (8 operand values: the caller chooses their distribution -- the sustained clock of this part depends on the operand VALUES, zeros
 * run ~19 % faster than N(0,1)); out: multiprocessors x 256 floats. FLOP per launch = 2 * 128*128*64 * iters * 4 * multiprocessors. */
int vt_probe_mfma(const uint16_t* a, const uint16_t* b, float* out, int iters, void* stream);
/* What a kernel that only READS sustains from HBM on THIS device (measurement aid for the roofline of the decode step, whose bytes are
 * weights and KV pages read once; ABI 113): `bytes` (>= 16, p 16-byte aligned) are read once, 16 bytes per lane and load; nt != 0 uses the
 * read-once cache policy of the weight-streaming GEMM. out: one word (never written in practice: it keeps the loads alive). */
int vt_probe_read(const void* p, size_t bytes, int nt, unsigned* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITRON_HIP_H */
