// vt_attn.hip -- attention kernels for gfx950.
//
//  * flash_attn_kernel<HD, CAUSAL>: tiled online-softmax attention on v_mfma_f32_32x32x16_bf16.
//      - ViT spatial attention (HD=64, no mask): CLIPAttention as called by the reference at
//        vitron/model/multimodal_encoder/languagebind/video/modeling_video.py:136-146
//      - LLaMA causal prefill (HD=128): LlamaAttention (transformers 4.31; restated in-repo at
//        reference vitron/train/llama_flash_attn_monkey_patch.py:30-66)
//    "All-swapped" formulation: S^T = K.Q^T and O^T = V^T.P^T, so a lane owns ONE query row
//    (lane&31) for the scores, the running max/sum and the output accumulator alike: the softmax
//    and the rescale are lane-local, P never leaves registers, and the only cross-lane traffic is
//    one lane<->lane+32 exchange per tile for the row max. K rows are fed to the MFMA through the
//    permutation pi(i) = 16*((i>>2)&1) + (i&3) + 4*(i>>3) so that each lane's 16 score registers are
//    16 consecutive keys -- which makes the P fragment of the second MFMA a plain bf16 pack, with V
//    stored transposed ([HD][64 keys] tiles) so its fragments are single ds_read_b128.
//    K / V^T live in 64-key tiles (= KV-cache pages) addressed through a tile table; tiles are staged
//    by LDS-DMA (global_load_lds_dwordx4) into a 2-stage LDS ring with XOR-swizzled chunks.
//  * kv_tiles_kernel<HD, ROPE>: fused-QKV rows -> (rotary embedding on q,k) -> K tiles / V^T tiles.
//  * attn_decode_fused_kernel<HD, ROPE>: one decode step's attention in one launch (rotary on the new q / k, k / v append to
//    the pages, single-query attention, in-block combine); attn_decode_kernel + _combine: the same on an existing cache as a
//    standalone op (split over the KV tiles). HBM-bound.
//  * attn_temporal8_kernel / attn_temporal_kernel<T>: the video tower's attention over T frames at one token position
//    (reference modeling_video.py:105-127) -- one wavefront per (clip, position, head).
#include <stdlib.h>

#include <atomic>

#include "vt_common.h"
#include "vt_kernels.h"
#include "vt_mx4.h"

namespace {

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

// chunk swizzles (16-B chunk index inside an LDS row) -- see file header
template <int HD>
__device__ __forceinline__ int k_swz(int key) {
  return (HD == 128) ? (key & 15) : ((key >> 1) & 7);
}
__device__ __forceinline__ int v_swz(int d) { return (d >> 1) & 7; }

#define RESCALE_THR 8.0f
// The softmax weights feed the P.V MFMA as fp16 (11 mantissa bits; V^T pages hold fp16, see vt_common.h). They are computed as
// 2^(s - m_run + P_BIAS): with the deferred rescale P <= 2^(RESCALE_THR + P_BIAS) = 2^15 < 65504, and the smallest NORMAL fp16
// (2^-14) sits 2^-21 below the running maximum's weight instead of 2^-14, so that the long tail of a peaked row is not lost to
// fp16's narrow exponent range. The row sum l carries the same factor, which cancels in O / l.
#define P_BIAS 7.0f

// NWAVES waves x 32 query rows share one K / V^T tile ring of NSTAGES 64-key tiles (LDS-DMA, counted vmcnt):
//   <4 waves, 2 stages>  128-row blocks, 64 KiB LDS (HD=128), two blocks per CU      -- ViT (many short sequences)
//   <8 waves, 3 stages>  256-row blocks, 96 KiB LDS, one block per CU, two tiles of DMA lead -- long causal prefill
// ABL (timing ablations, results are garbage): bit0 = no exp / max / sum (softmax VALU), bit1 = no MFMA;
// experiment knobs (results correct): bit2 = s_setprio 3 around the softmax section, bit3 = s_setprio 3 around the MFMA sections
// MXOUT (precise level 3, head_dim 128): the epilogue also writes the MX-FP4 image of the output's rounding remainder (O4 / oexp: o_proj's
// second operand, vt_mx4.h); the other instantiations never read the two pointers.
template <int HD, bool CAUSAL, int NWAVES, int NSTAGES, int ABL = 0, bool MXOUT = false>
__global__ __launch_bounds__(64 * NWAVES, (NWAVES == 8) ? 2 : 2) void flash_attn_kernel(const bf16_t* __restrict__ Q, int ldq,
                                                         const bf16_t* __restrict__ Kt,
                                                         const bf16_t* __restrict__ Vt,
                                                         const int* __restrict__ tile_table,
                                                         const VtAttnSeq* __restrict__ seqs, bf16_t* __restrict__ O,
                                                         int ldo, int heads, float scale_log2e, uint8_t* __restrict__ O4,
                                                         uint8_t* __restrict__ oexp) {
  constexpr int KS = HD / 16;          // k-steps of the QK^T MFMA
  constexpr int DB = HD / 32;          // 32-wide d blocks of the output
  constexpr int KROW = HD * 2;         // bytes per K row
  constexpr int TILE_BYTES = 64 * HD * 2;  // K tile == V^T tile == 64*HD bf16
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  constexpr int PIECES = TILE_BYTES / 1024;  // 1-KiB DMA pieces per tile
  constexpr int PPW = PIECES / NWAVES;       // per wave
  constexpr int QBLK = 32 * NWAVES;          // query rows per workgroup
  static_assert(PIECES % NWAVES == 0, "tile pieces must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const VtAttnSeq sq = seqs[blockIdx.z];
  const int nqb = (sq.q_len + QBLK - 1) / QBLK;
  // grid = (heads, q blocks, sequences): workgroups are dispatched x-fastest, so ALL heads' heaviest (latest) causal
  // blocks start first and the light ones fill the tail (longest-processing-time order)
  const int qb = nqb - 1 - (int)blockIdx.y;
  if (qb < 0) return;
  const int head = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ql = lane & 31, hh = lane >> 5;
  const int past = sq.kv_len - sq.q_len;
  const int q0 = qb * QBLK;
  const int qrow = q0 + wave * 32 + ql;             // row inside the sequence (may be >= q_len)
  const int qrow_c = min(qrow, sq.q_len - 1);

  // number of 64-key tiles this block needs
  int ntiles = (sq.kv_len + 63) >> 6;
  if (CAUSAL) {
    const int last_key = past + min(q0 + QBLK - 1, sq.q_len - 1);
    ntiles = min(ntiles, (last_key >> 6) + 1);
  }

  // ---- Q fragments (B operand): lane (q, h) holds d = ks*16 + h*8 .. +7 ---------------------------
  bf16x8 qf[KS];
  {
    const bf16_t* qp = Q + (size_t)(sq.q_row0 + qrow_c) * ldq + head * HD + hh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }

  // ---- DMA source offsets (elements inside a tile), swizzle applied on the source side -------------
  // K tile piece p covers LDS bytes [p*1024, +1024): rows of KROW bytes.
  int k_src_off[PPW], v_src_off[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int byte = (wave * PPW + i) * 1024 + lane * 16;
    {
      const int row = byte / KROW, c = (byte % KROW) >> 4;
      k_src_off[i] = row * HD + ((c ^ k_swz<HD>(row)) << 3);
    }
    {
      const int row = byte >> 7, c = (byte & 127) >> 4;  // V^T rows are 64 keys = 128 B
      v_src_off[i] = row * 64 + ((c ^ v_swz(row)) << 3);
    }
  }
  const size_t head_off = (size_t)head * 64 * HD;
  const size_t tile_stride = (size_t)heads * 64 * HD;
  const int* table = tile_table + sq.table_off;

  auto stage = [&](int buf, int t) {
    const size_t toff = (size_t)table[t] * tile_stride + head_off;
    char* base = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16(Kt + toff + k_src_off[i], base + (wave * PPW + i) * 1024);
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16(Vt + toff + v_src_off[i], base + TILE_BYTES + (wave * PPW + i) * 1024);
  };

  // ---- fragment read offsets ---------------------------------------------------------------------------
  // K: MFMA row i = lane&31 reads key pi(i) of the 32-key sub tile, chunk ks*2 + h
  const int pi = 16 * ((ql >> 2) & 1) + (ql & 3) + 4 * (ql >> 3);
  int k_row_off[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) k_row_off[sub] = (sub * 32 + pi) * KROW;
  const int k_sw0 = k_swz<HD>(pi);  // same for sub 0/1 (32 is a multiple of both swizzle periods)
  // V^T: MFMA row i = lane&31 reads d = db*32 + i, chunk sub*4 + 2h + j
  const int v_sw = v_swz(ql);       // (db*32 + ql) has the same swizzle as ql

  f32x16 oacc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // counted wait: at the end of iteration t (prologue = t -1) tile t+1 must have landed; the stages of tiles
  // t+2 .. t+NSTAGES-1 (PPW K pieces + PPW V pieces each) may stay in flight across the barrier
#define FA_TILE_SYNC(T)                                                                              \
  do {                                                                                               \
    if ((NSTAGES == 3) && ((T) + 2 < ntiles)) {                                                      \
      if constexpr (PPW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                       \
      else if constexpr (PPW == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                  \
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                          \
    } else {                                                                                         \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
    }                                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
    __builtin_amdgcn_s_barrier();                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  } while (0)

#pragma unroll
  for (int s0 = 0; s0 < NSTAGES - 1; ++s0)
    if (s0 < ntiles) stage(s0, s0);
  FA_TILE_SYNC(-1);

  int slot = 0, pre_slot = NSTAGES - 1;
  for (int t = 0; t < ntiles; ++t) {
    if (t + NSTAGES - 1 < ntiles) stage(pre_slot, t + NSTAGES - 1);
    const char* kb = smem + slot * STAGE_BYTES;
    const char* vb = kb + TILE_BYTES;

    // ---- S^T = K . Q^T ---------------------------------------------------------------------------------
    if (ABL & 8) __builtin_amdgcn_s_setprio(3);
    // (the two 32-key sub tiles alternate so consecutive MFMAs never depend on each other: a 32x32x16 result is
    //  only available ~64 cycles after issue, twice its 32-cycle issue slot)
    f32x16 sacc[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[sub][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int chunk = (ks * 2 + hh) ^ k_sw0;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const bf16x8 kf = *(const bf16x8*)(kb + k_row_off[sub] + (chunk << 4));
        if (!(ABL & 2)) sacc[sub] = VT_MFMA_32x32x16(kf, qf[ks], sacc[sub]);
        else sacc[sub][ks] += __builtin_bit_cast(float, (uint32_t)kf[0] << 16);
      }
    }
    // lane (q,h): sacc[sub][r] = S[q][t*64 + sub*32 + 16h + r]

    if (ABL & 8) __builtin_amdgcn_s_setprio(0);
    if (ABL & 4) __builtin_amdgcn_s_setprio(3);
    // ---- mask (only tiles that touch the diagonal or the end of the keys) ----------------------------------
    const int key0 = t * 64;
    const bool need_mask = (key0 + 64 > sq.kv_len) || (CAUSAL && (key0 + 63 > past + q0 + wave * 32));
    if (need_mask) {
      // a REAL branch: without the empty asm the compiler if-converts this block into 3 VALU instructions per score on EVERY
      // tile (92 of the loop's 240 VALU instructions, ISA dump) although only the diagonal / last tiles need them
      asm volatile("" ::: "memory");
      const int lim = CAUSAL ? min(sq.kv_len - 1, past + qrow) : (sq.kv_len - 1);  // last visible key
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + sub * 32 + 16 * hh + r;
          if (key > lim) sacc[sub][r] = -INFINITY;
        }
    }

    // ---- online softmax (log2 domain, deferred rescale) ------------------------------------------------------------
    // The running max m_run only moves when some row's tile max exceeds it by more than RESCALE_THR (wave-uniform
    // branch): P is then bounded by 2^RESCALE_THR instead of 1, which bf16's exponent range absorbs at unchanged
    // relative precision, and the O / l rescale (the expensive VALU part) is skipped on almost every tile.
    float mx = fmaxf(sacc[0][0], sacc[1][0]);
    if (!(ABL & 1)) {
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = __builtin_fmaxf(__builtin_fmaxf(mx, sacc[0][r]), sacc[1][r]);   // v_max3_f32
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    }
    const float m_tile = mx * scale_log2e;
    if (__builtin_amdgcn_ballot_w64(m_tile > m_run + RESCALE_THR) != 0) {
      const float m_new = fmaxf(m_run, m_tile);
      const float alpha = (m_new == -INFINITY) ? 1.f : fast_exp2(m_run - m_new);  // m_run = -inf -> 0
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    }
    const float m_safe = (m_run == -INFINITY) ? 0.f : m_run;
    float psum = 0.f;
    f16x8 pf[2][2];
    // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32: two lanes' worth per instruction) for the scale-and-shift and the row sum:
    // the softmax VALU work, not the MFMAs, is what fills the SIMD here (PMC: VALU busy 45 %, MFMA busy 42 %, 16 % co-issue)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 sc2 = {scale_log2e, scale_log2e}, ms2 = {P_BIAS - m_safe, P_BIAS - m_safe};
    f32x2 psum2 = {0.f, 0.f};
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        if (!(ABL & 1)) {
          const f32x2 a = {sacc[sub][r], sacc[sub][r + 1]};
          const f32x2 y = __builtin_elementwise_fma(a, sc2, ms2);
          f32x2 e;
          e.x = fast_exp2(y.x);
          e.y = fast_exp2(y.y);
          psum2 += e;
          sacc[sub][r] = e.x;
          sacc[sub][r + 1] = e.y;
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        u32x4 w;
        w.x = pack_f16x2(sacc[sub][8 * j + 0], sacc[sub][8 * j + 1]);
        w.y = pack_f16x2(sacc[sub][8 * j + 2], sacc[sub][8 * j + 3]);
        w.z = pack_f16x2(sacc[sub][8 * j + 4], sacc[sub][8 * j + 5]);
        w.w = pack_f16x2(sacc[sub][8 * j + 6], sacc[sub][8 * j + 7]);
        pf[sub][j] = __builtin_bit_cast(f16x8, w);
      }
    }
    psum = psum2.x + psum2.y;
    l_run += psum;

    if (ABL & 4) __builtin_amdgcn_s_setprio(0);
    if (ABL & 8) __builtin_amdgcn_s_setprio(3);
    // ---- O^T += V^T . P^T --------------------------------------------------------------------------------------
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int chunk = (sub * 4 + 2 * hh + j) ^ v_sw;
#pragma unroll
        for (int db = 0; db < DB; ++db) {   // DB independent accumulators back to back
          const f16x8 vf = *(const f16x8*)(vb + (db * 32 + ql) * 128 + (chunk << 4));
          if (!(ABL & 2)) oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[sub][j], oacc[db], 0, 0, 0);
          else oacc[db][j] += (float)vf[0];
        }
      }
    if (ABL & 8) __builtin_amdgcn_s_setprio(0);
    FA_TILE_SYNC(t);
    slot = (slot + 1 == NSTAGES) ? 0 : slot + 1;
    pre_slot = (pre_slot + 1 == NSTAGES) ? 0 : pre_slot + 1;
  }
#undef FA_TILE_SYNC

  // ---- epilogue ------------------------------------------------------------------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = (l_tot > 0.f) ? 1.f / l_tot : 0.f;
  if constexpr (MXOUT) {
    const bool valid = qrow < sq.q_len;
    const int m = sq.q_row0 + qrow_c;
    bf16_t* op = O + (size_t)m * ldo + head * HD + 4 * hh;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      float lo[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float a0 = oacc[db][4 * g + 0] * inv, a1 = oacc[db][4 * g + 1] * inv, a2 = oacc[db][4 * g + 2] * inv, a3 = oacc[db][4 * g + 3] * inv;
        u32x2 o;
        o.x = pack_op2(a0, a1);
        o.y = pack_op2(a2, a3);
        if (valid) *(u32x2*)(op + db * 32 + 8 * g) = o;
        lo[4 * g + 0] = a0 - oplo_to_f32(o.x);
        lo[4 * g + 1] = a1 - ophi_to_f32(o.x);
        lo[4 * g + 2] = a2 - oplo_to_f32(o.y);
        lo[4 * g + 3] = a3 - ophi_to_f32(o.y);
      }
      mx4_store_attn_block(lo, hh, valid, O4 + (size_t)m * (heads * (HD / 2)) + head * (HD / 2) + db * 16,
                           oexp + mx4_aexp_index(m, head * DB + db, heads * DB));
    }
  } else if (qrow < sq.q_len) {
    bf16_t* op = O + (size_t)(sq.q_row0 + qrow) * ldo + head * HD + 4 * hh;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 o;
        o.x = pack_op2(oacc[db][4 * g + 0] * inv, oacc[db][4 * g + 1] * inv);
        o.y = pack_op2(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv);
        *(u32x2*)(op + db * 32 + 8 * g) = o;
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// flash_attn_precise_kernel (vt_llama_model.precise_qk: head_dim 128, causal; vt_vit_model.precise: head_dim 64, the towers' spatial
// attention; 4 waves x 32 query rows, 2-stage ring): flash_attn_kernel's
// formulation with q and k as operand PAIRS written by kv_tiles_precise_kernel -- S^T = K_hi.(Q_hi + Q_lo)^T + K_lo.Q_hi^T, three MFMAs
// per k step instead of one (the K_lo.Q_lo term is 2^-22 of the score and dropped): the score carries ~2^-20 of operand
// rounding instead of 2^-12 (fp16) / 2^-9 (bf16), which is what the softmax amplifies. K_lo exists for the keys this
// pass appended (a per-pass workspace: tile `t - (past >> 6)` of the sequence); older keys (prefix reuse, chunked prefill) have
// K_hi only. P, V^T and the P.V MFMA are unchanged. LDS: K_hi + K_lo + V^T = 48 KiB per stage, 96 KiB: one workgroup per CU.
// ------------------------------------------------------------------------------------------------------------------
template <bool CAUSAL, int HD>
__global__ __launch_bounds__(256, 1) void flash_attn_precise_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ Qlo,
                                                                    int ldqlo, const bf16_t* __restrict__ Kt,
                                                                    const bf16_t* __restrict__ Klo, int klo_tiles_per_seq,
                                                                    const bf16_t* __restrict__ Vt, const int* __restrict__ tile_table,
                                                                    const VtAttnSeq* __restrict__ seqs, bf16_t* __restrict__ O, int ldo,
                                                                    bf16_t* __restrict__ Olo, int heads, float scale_log2e) {
  constexpr int NWAVES = 4;
  constexpr int KS = HD / 16, DB = HD / 32, KROW = HD * 2, TILE_BYTES = 64 * HD * 2, STAGE_BYTES = 3 * TILE_BYTES;
  constexpr int PIECES = TILE_BYTES / 1024, PPW = PIECES / NWAVES, QBLK = 32 * NWAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const VtAttnSeq sq = seqs[blockIdx.z];
  const int nqb = (sq.q_len + QBLK - 1) / QBLK;
  const int qb = nqb - 1 - (int)blockIdx.y;
  if (qb < 0) return;
  const int head = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ql = lane & 31, hh = lane >> 5;
  const int past = sq.kv_len - sq.q_len;
  const int q0 = qb * QBLK;
  const int qrow = q0 + wave * 32 + ql;
  const int qrow_c = min(qrow, sq.q_len - 1);
  int ntiles = (sq.kv_len + 63) >> 6;
  if (CAUSAL) {
    const int last_key = past + min(q0 + QBLK - 1, sq.q_len - 1);
    ntiles = min(ntiles, (last_key >> 6) + 1);
  }
  const int t_new0 = past >> 6;    // first tile that holds a key of this pass: K_lo exists from here on

  bf16x8 qf[KS], qfl[KS];
  {
    const bf16_t* qp = Q + (size_t)(sq.q_row0 + qrow_c) * ldq + head * HD + hh * 8;
    const bf16_t* qlp = Qlo + (size_t)(sq.q_row0 + qrow_c) * ldqlo + head * HD + hh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = *(const bf16x8*)(qp + ks * 16);
      qfl[ks] = *(const bf16x8*)(qlp + ks * 16);
    }
  }
  int k_src_off[PPW], v_src_off[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int byte = (wave * PPW + i) * 1024 + lane * 16;
    {
      const int row = byte / KROW, c = (byte % KROW) >> 4;
      k_src_off[i] = row * HD + ((c ^ k_swz<HD>(row)) << 3);
    }
    {
      const int row = byte >> 7, c = (byte & 127) >> 4;
      v_src_off[i] = row * 64 + ((c ^ v_swz(row)) << 3);
    }
  }
  const size_t head_off = (size_t)head * 64 * HD;
  const size_t tile_stride = (size_t)heads * 64 * HD;
  const int* table = tile_table + sq.table_off;
  const bf16_t* klo_seq = Klo + (size_t)blockIdx.z * klo_tiles_per_seq * tile_stride + head_off;

  auto stage = [&](int buf, int t) {
    const size_t toff = (size_t)table[t] * tile_stride + head_off;
    char* base = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16(Kt + toff + k_src_off[i], base + (wave * PPW + i) * 1024);
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16(Vt + toff + v_src_off[i], base + TILE_BYTES + (wave * PPW + i) * 1024);
    if (t >= t_new0) {
      const bf16_t* kl = klo_seq + (size_t)(t - t_new0) * tile_stride;
#pragma unroll
      for (int i = 0; i < PPW; ++i) glds16(kl + k_src_off[i], base + 2 * TILE_BYTES + (wave * PPW + i) * 1024);
    }
  };

  const int pi = 16 * ((ql >> 2) & 1) + (ql & 3) + 4 * (ql >> 3);
  int k_row_off[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) k_row_off[sub] = (sub * 32 + pi) * KROW;
  const int k_sw0 = k_swz<HD>(pi);
  const int v_sw = v_swz(ql);

  f32x16 oacc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

#define FAP_TILE_SYNC()                                  \
  do {                                                   \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                        \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)

  if (ntiles > 0) stage(0, 0);
  FAP_TILE_SYNC();
  int slot = 0;
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) stage(slot ^ 1, t + 1);
    const char* kb = smem + slot * STAGE_BYTES;
    const char* vb = kb + TILE_BYTES;
    const char* klb = kb + 2 * TILE_BYTES;
    const bool has_lo = t >= t_new0;     // block-uniform

    f32x16 sacc[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[sub][r] = 0.f;
    // small terms first (K_hi.Q_lo, K_lo.Q_hi), the dominant K_hi.Q_hi last: the fp32 accumulator keeps the small terms' low bits
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int chunk = (ks * 2 + hh) ^ k_sw0;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const bf16x8 kf = *(const bf16x8*)(kb + k_row_off[sub] + (chunk << 4));
        sacc[sub] = VT_MFMA_32x32x16(kf, qfl[ks], sacc[sub]);
      }
    }
    if (has_lo) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int chunk = (ks * 2 + hh) ^ k_sw0;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const bf16x8 kl = *(const bf16x8*)(klb + k_row_off[sub] + (chunk << 4));
          sacc[sub] = VT_MFMA_32x32x16(kl, qf[ks], sacc[sub]);
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int chunk = (ks * 2 + hh) ^ k_sw0;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const bf16x8 kf = *(const bf16x8*)(kb + k_row_off[sub] + (chunk << 4));
        sacc[sub] = VT_MFMA_32x32x16(kf, qf[ks], sacc[sub]);
      }
    }

    const int key0 = t * 64;
    const bool need_mask = (key0 + 64 > sq.kv_len) || (CAUSAL && (key0 + 63 > past + q0 + wave * 32));
    if (need_mask) {
      asm volatile("" ::: "memory");
      const int lim = CAUSAL ? min(sq.kv_len - 1, past + qrow) : (sq.kv_len - 1);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + sub * 32 + 16 * hh + r;
          if (key > lim) sacc[sub][r] = -INFINITY;
        }
    }
    float mx = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = __builtin_fmaxf(__builtin_fmaxf(mx, sacc[0][r]), sacc[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_tile = mx * scale_log2e;
    if (__builtin_amdgcn_ballot_w64(m_tile > m_run + RESCALE_THR) != 0) {
      const float m_new = fmaxf(m_run, m_tile);
      const float alpha = (m_new == -INFINITY) ? 1.f : fast_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    }
    const float m_safe = (m_run == -INFINITY) ? 0.f : m_run;
    f16x8 pf[2][2];
    float psum = 0.f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = fast_exp2(__builtin_fmaf(sacc[sub][r], scale_log2e, P_BIAS - m_safe));
        psum += e;
        sacc[sub][r] = e;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        u32x4 w;
        w.x = pack_f16x2(sacc[sub][8 * j + 0], sacc[sub][8 * j + 1]);
        w.y = pack_f16x2(sacc[sub][8 * j + 2], sacc[sub][8 * j + 3]);
        w.z = pack_f16x2(sacc[sub][8 * j + 4], sacc[sub][8 * j + 5]);
        w.w = pack_f16x2(sacc[sub][8 * j + 6], sacc[sub][8 * j + 7]);
        pf[sub][j] = __builtin_bit_cast(f16x8, w);
      }
    }
    l_run += psum;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int chunk = (sub * 4 + 2 * hh + j) ^ v_sw;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const f16x8 vf = *(const f16x8*)(vb + (db * 32 + ql) * 128 + (chunk << 4));
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[sub][j], oacc[db], 0, 0, 0);
        }
      }
    FAP_TILE_SYNC();
    slot ^= 1;
  }
#undef FAP_TILE_SYNC

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = (l_tot > 0.f) ? 1.f / l_tot : 0.f;
  if (qrow < sq.q_len) {
    bf16_t* op = O + (size_t)(sq.q_row0 + qrow) * ldo + head * HD + 4 * hh;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float a0 = oacc[db][4 * g + 0] * inv, a1 = oacc[db][4 * g + 1] * inv, a2 = oacc[db][4 * g + 2] * inv, a3 = oacc[db][4 * g + 3] * inv;
        u32x2 o;
        o.x = pack_op2(a0, a1);
        o.y = pack_op2(a2, a3);
        *(u32x2*)(op + db * 32 + 8 * g) = o;
        if (Olo) {   // level 2: the attention output as an operand pair (o_proj's A operand)
          u32x2 l;
          l.x = pack_op2(a0 - oplo_to_f32(o.x), a1 - ophi_to_f32(o.x));
          l.y = pack_op2(a2 - oplo_to_f32(o.y), a3 - ophi_to_f32(o.y));
          *(u32x2*)(Olo + (size_t)(sq.q_row0 + qrow) * ldo + head * HD + 4 * hh + db * 32 + 8 * g) = l;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// kv_tiles_kernel: one block per (64-position tile, head, sequence). Handles the NEW tokens of the sequence
// (positions past .. kv_len-1) that fall into the tile:
//   K tile [64][HD]: row = position & 63   (rotary applied when ROPE)
//   V^T tile [HD][64]: column = position & 63
//   q rows rotated in place when ROPE.
// A tile whose first position is new is written completely (padding rows/cols zero-filled) so the flash
// kernel never sees non-finite padding.
// ------------------------------------------------------------------------------------------------------------------
template <int HD, bool ROPE>
__global__ __launch_bounds__(256) void kv_tiles_kernel(bf16_t* __restrict__ qkv, int ldqkv, int q_col0, int k_col0,
                                                       int v_col0, bf16_t* __restrict__ Kt, bf16_t* __restrict__ Vt,
                                                       const int* __restrict__ tile_table,
                                                       const VtAttnSeq* __restrict__ seqs, int heads,
                                                       const float* __restrict__ rope_cos,
                                                       const float* __restrict__ rope_sin,
                                                       const int* __restrict__ positions) {
  constexpr int CH = HD / 8;  // 16-B chunks per row
  __shared__ __attribute__((aligned(16))) bf16_t vs[64][HD + 8];  // V rows of this tile (padded)
  const VtAttnSeq sq = seqs[blockIdx.z];
  const int past = sq.kv_len - sq.q_len;
  const int t = (past >> 6) + blockIdx.x;  // absolute tile index
  if (t * 64 >= sq.kv_len) return;
  const int head = blockIdx.y;
  const int p_lo = max(past, t * 64), p_hi = min(sq.kv_len, t * 64 + 64);  // new positions [p_lo, p_hi)
  const bool fresh = (t * 64 >= past);  // tile starts with a new position -> full write
  const size_t toff = ((size_t)tile_table[sq.table_off + t] * heads + head) * 64 * HD;
  bf16_t* kt = Kt + toff;
  bf16_t* vt = Vt + toff;

  // ---- K rows (+ rope) and Q rope: thread -> (row, chunk pair) ------------------------------------------------
  for (int it = threadIdx.x; it < 64 * (CH / 2); it += 256) {
    const int r = it / (CH / 2), c = it % (CH / 2);  // row in tile, chunk in first half of the head dim
    const int pos = t * 64 + r;
    if (pos >= p_lo && pos < p_hi) {
      const int row = sq.q_row0 + (pos - past);
      bf16_t* kp = qkv + (size_t)row * ldqkv + k_col0 + head * HD;
      u32x4 lo = *(const u32x4*)(kp + c * 8);
      u32x4 hi = *(const u32x4*)(kp + HD / 2 + c * 8);
      if (ROPE) {
        const int rp = positions[row];
        const float* cs = rope_cos + (size_t)rp * (HD / 2) + c * 8;
        const float* sn = rope_sin + (size_t)rp * (HD / 2) + c * 8;
        bf16_t* qp = qkv + (size_t)row * ldqkv + q_col0 + head * HD;
        u32x4 qlo = *(const u32x4*)(qp + c * 8);
        u32x4 qhi = *(const u32x4*)(qp + HD / 2 + c * 8);
        u32x4 klo_o, khi_o, qlo_o, qhi_o;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float c0 = cs[2 * w], c1 = cs[2 * w + 1], s0 = sn[2 * w], s1 = sn[2 * w + 1];
          {
            const float a0 = oplo_to_f32(lo[w]), a1 = ophi_to_f32(lo[w]);
            const float b0 = oplo_to_f32(hi[w]), b1 = ophi_to_f32(hi[w]);
            klo_o[w] = pack_op2(rope_lo(a0, b0, c0, s0), rope_lo(a1, b1, c1, s1));
            khi_o[w] = pack_op2(rope_hi(a0, b0, c0, s0), rope_hi(a1, b1, c1, s1));
          }
          {
            const float a0 = oplo_to_f32(qlo[w]), a1 = ophi_to_f32(qlo[w]);
            const float b0 = oplo_to_f32(qhi[w]), b1 = ophi_to_f32(qhi[w]);
            qlo_o[w] = pack_op2(rope_lo(a0, b0, c0, s0), rope_lo(a1, b1, c1, s1));
            qhi_o[w] = pack_op2(rope_hi(a0, b0, c0, s0), rope_hi(a1, b1, c1, s1));
          }
        }
        lo = klo_o;
        hi = khi_o;
        *(u32x4*)(qp + c * 8) = qlo_o;
        *(u32x4*)(qp + HD / 2 + c * 8) = qhi_o;
      }
      *(u32x4*)(kt + r * HD + c * 8) = lo;
      *(u32x4*)(kt + r * HD + HD / 2 + c * 8) = hi;
    } else if (fresh && pos >= p_hi) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      *(u32x4*)(kt + r * HD + c * 8) = z;
      *(u32x4*)(kt + r * HD + HD / 2 + c * 8) = z;
    }
  }

  // ---- V rows -> LDS -> V^T ----------------------------------------------------------------------------------------
  for (int it = threadIdx.x; it < 64 * CH; it += 256) {
    const int r = it / CH, c = it % CH;
    const int pos = t * 64 + r;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (pos >= p_lo && pos < p_hi) {
      const int row = sq.q_row0 + (pos - past);
      v = *(const u32x4*)(qkv + (size_t)row * ldqkv + v_col0 + head * HD + c * 8);
    }
    *(u32x4*)(&vs[r][c * 8]) = v;
  }
  __syncthreads();
  // thread -> (d, 8-key chunk)
  for (int it = threadIdx.x; it < HD * 8; it += 256) {
    const int d = it >> 3, kc = it & 7;
    const int pos0 = t * 64 + kc * 8;
    // fresh tile: every chunk is ours (new keys then zero padding). Otherwise only chunks with new keys,
    // and a chunk that also holds old keys / old padding is written element-wise.
    const bool full = fresh || (pos0 >= p_lo && pos0 + 8 <= p_hi);
    const bool any_new = (pos0 < p_hi) && (pos0 + 8 > p_lo);
    if (!full && !any_new) continue;
    float vals[8];   // bf16 -> fp16 (exact in fp16's normal range, saturating): the V^T pages hold fp16, see vt_common.h
#pragma unroll
    for (int j = 0; j < 8; ++j) vals[j] = vt_clamp_f16(op_to_f32(vs[kc * 8 + j][d]));
    uint16_t* dst = vt + d * 64 + kc * 8;
    u32x4 w;
    w.x = pack_f16x2(vals[0], vals[1]);
    w.y = pack_f16x2(vals[2], vals[3]);
    w.z = pack_f16x2(vals[4], vals[5]);
    w.w = pack_f16x2(vals[6], vals[7]);
    if (full) {
      *(u32x4*)dst = w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int pos = pos0 + j;
        if (pos >= p_lo && pos < p_hi) dst[j] = (uint16_t)((w[j >> 1] >> (16 * (j & 1))) & 0xffffu);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// kv_tiles_precise_kernel (vt_llama_model.precise_qk): the same job as kv_tiles_kernel<HD, true> for a q / k projection that
// arrives in FP32 (qk32 [rows][2 * heads * HD]: q columns, then k columns): the rotary embedding is applied in fp32 and q / k are
// rounded ONCE, each to an operand PAIR (hi = op(v), lo = op(v - f32(hi))):
//   q_hi -> the q columns of the fused qkv buffer (where the attention kernel reads Q), q_lo -> qlo [rows][heads * HD]
//   k_hi -> the K page (what decode steps and later passes see: the ordinary cache), k_lo -> klo, a per-pass workspace of
//           [sequence][new tile][head][64][HD] (zero rows where the tile holds no new key)
// V rows go to the V^T pages exactly as in kv_tiles_kernel. flash_attn_kernel<..., PREC = true> consumes the pairs.
// ------------------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void kv_tiles_precise_kernel(const float* __restrict__ qk32, int ld32, int v_f32,
                                                               bf16_t* __restrict__ qkv, int ldqkv,
                                                               int q_col0, int v_col0, bf16_t* __restrict__ qlo,
                                                               bf16_t* __restrict__ Kt, bf16_t* __restrict__ Vt, bf16_t* __restrict__ klo,
                                                               const int* __restrict__ tile_table, const VtAttnSeq* __restrict__ seqs,
                                                               int heads, const float* __restrict__ rope_cos,
                                                               const float* __restrict__ rope_sin, const int* __restrict__ positions) {
  constexpr int CH = HD / 8;
  __shared__ __attribute__((aligned(16))) bf16_t vs[64][HD + 8];
  const VtAttnSeq sq = seqs[blockIdx.z];
  const int past = sq.kv_len - sq.q_len;
  const int t = (past >> 6) + blockIdx.x;
  if (t * 64 >= sq.kv_len) return;
  const int head = blockIdx.y;
  const int H = heads * HD;
  const int p_lo = max(past, t * 64), p_hi = min(sq.kv_len, t * 64 + 64);
  const bool fresh = (t * 64 >= past);
  const size_t toff = ((size_t)tile_table[sq.table_off + t] * heads + head) * 64 * HD;
  bf16_t* kt = Kt + toff;
  bf16_t* vt = Vt + toff;
  bf16_t* kl = klo + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * heads + head) * 64 * HD;

  for (int it = threadIdx.x; it < 64 * (CH / 2); it += 256) {
    const int r = it / (CH / 2), c = it % (CH / 2);
    const int pos = t * 64 + r;
    u32x4 k_h_lo = {0u, 0u, 0u, 0u}, k_h_hi = k_h_lo, k_l_lo = k_h_lo, k_l_hi = k_h_lo;
    const bool is_new = pos >= p_lo && pos < p_hi;
    if (is_new) {
      const int row = sq.q_row0 + (pos - past);
      const bool rope = rope_cos != nullptr;     // the towers have no rotary embedding: identity (a * 1 - b * 0 is exact)
      const int rp = rope ? positions[row] : 0;
      const float* cs = rope_cos + (size_t)rp * (HD / 2) + c * 8;
      const float* sn = rope_sin + (size_t)rp * (HD / 2) + c * 8;
      const float* qs = qk32 + (size_t)row * ld32 + head * HD;
      const float* ks = qs + H;
      u32x4 q_h_lo, q_h_hi, q_l_lo, q_l_hi;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float c0 = rope ? cs[2 * w] : 1.f, c1 = rope ? cs[2 * w + 1] : 1.f, s0 = rope ? sn[2 * w] : 0.f, s1 = rope ? sn[2 * w + 1] : 0.f;
        {
          const float a0 = ks[c * 8 + 2 * w], a1 = ks[c * 8 + 2 * w + 1];
          const float b0 = ks[HD / 2 + c * 8 + 2 * w], b1 = ks[HD / 2 + c * 8 + 2 * w + 1];
          const float lo0 = rope_lo(a0, b0, c0, s0), lo1 = rope_lo(a1, b1, c1, s1);
          const float hi0 = rope_hi(a0, b0, c0, s0), hi1 = rope_hi(a1, b1, c1, s1);
          k_h_lo[w] = pack_op2(lo0, lo1);
          k_h_hi[w] = pack_op2(hi0, hi1);
          k_l_lo[w] = pack_op2(lo0 - oplo_to_f32(k_h_lo[w]), lo1 - ophi_to_f32(k_h_lo[w]));
          k_l_hi[w] = pack_op2(hi0 - oplo_to_f32(k_h_hi[w]), hi1 - ophi_to_f32(k_h_hi[w]));
        }
        {
          const float a0 = qs[c * 8 + 2 * w], a1 = qs[c * 8 + 2 * w + 1];
          const float b0 = qs[HD / 2 + c * 8 + 2 * w], b1 = qs[HD / 2 + c * 8 + 2 * w + 1];
          const float lo0 = rope_lo(a0, b0, c0, s0), lo1 = rope_lo(a1, b1, c1, s1);
          const float hi0 = rope_hi(a0, b0, c0, s0), hi1 = rope_hi(a1, b1, c1, s1);
          q_h_lo[w] = pack_op2(lo0, lo1);
          q_h_hi[w] = pack_op2(hi0, hi1);
          q_l_lo[w] = pack_op2(lo0 - oplo_to_f32(q_h_lo[w]), lo1 - ophi_to_f32(q_h_lo[w]));
          q_l_hi[w] = pack_op2(hi0 - oplo_to_f32(q_h_hi[w]), hi1 - ophi_to_f32(q_h_hi[w]));
        }
      }
      bf16_t* qp = qkv + (size_t)row * ldqkv + q_col0 + head * HD;
      *(u32x4*)(qp + c * 8) = q_h_lo;
      *(u32x4*)(qp + HD / 2 + c * 8) = q_h_hi;
      bf16_t* qlp = qlo + (size_t)row * H + head * HD;
      *(u32x4*)(qlp + c * 8) = q_l_lo;
      *(u32x4*)(qlp + HD / 2 + c * 8) = q_l_hi;
    }
    if (is_new || (fresh && pos >= p_hi)) {
      *(u32x4*)(kt + r * HD + c * 8) = k_h_lo;
      *(u32x4*)(kt + r * HD + HD / 2 + c * 8) = k_h_hi;
    }
    *(u32x4*)(kl + r * HD + c * 8) = k_l_lo;              // the workspace tile is written completely: zeros where no new key sits
    *(u32x4*)(kl + r * HD + HD / 2 + c * 8) = k_l_hi;
  }

  for (int it = threadIdx.x; it < 64 * CH; it += 256) {
    const int r = it / CH, c = it % CH;
    const int pos = t * 64 + r;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (pos >= p_lo && pos < p_hi) {
      const int row = sq.q_row0 + (pos - past);
      if (v_f32) {   // level 2: v arrives in fp32 behind q | k (one rounding, here)
        const float* vp = qk32 + (size_t)row * ld32 + 2 * H + head * HD + c * 8;
        const f32x4 a = *(const f32x4*)vp, b = *(const f32x4*)(vp + 4);
        // straight into the pages' fp16 (both operand builds: the bf16 build would otherwise round v to 8 bits first)
        v.x = pack_f16x2(vt_clamp_f16(a[0]), vt_clamp_f16(a[1]));
        v.y = pack_f16x2(vt_clamp_f16(a[2]), vt_clamp_f16(a[3]));
        v.z = pack_f16x2(vt_clamp_f16(b[0]), vt_clamp_f16(b[1]));
        v.w = pack_f16x2(vt_clamp_f16(b[2]), vt_clamp_f16(b[3]));
      } else {
        v = *(const u32x4*)(qkv + (size_t)row * ldqkv + v_col0 + head * HD + c * 8);
      }
    }
    *(u32x4*)(&vs[r][c * 8]) = v;
  }
  __syncthreads();
  for (int it = threadIdx.x; it < HD * 8; it += 256) {
    const int d = it >> 3, kc = it & 7;
    const int pos0 = t * 64 + kc * 8;
    const bool full = fresh || (pos0 >= p_lo && pos0 + 8 <= p_hi);
    const bool any_new = (pos0 < p_hi) && (pos0 + 8 > p_lo);
    if (!full && !any_new) continue;
    uint16_t* dst = vt + d * 64 + kc * 8;
    u32x4 w;
    if (v_f32) {   // the staging buffer already holds fp16
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = (uint32_t)vs[kc * 8 + 2 * q][d] | ((uint32_t)vs[kc * 8 + 2 * q + 1][d] << 16);
    } else {
      float vals[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) vals[j] = vt_clamp_f16(op_to_f32(vs[kc * 8 + j][d]));
      w.x = pack_f16x2(vals[0], vals[1]);
      w.y = pack_f16x2(vals[2], vals[3]);
      w.z = pack_f16x2(vals[4], vals[5]);
      w.w = pack_f16x2(vals[6], vals[7]);
    }
    if (full) {
      *(u32x4*)dst = w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int pos = pos0 + j;
        if (pos >= p_lo && pos < p_hi) dst[j] = (uint16_t)((w[j >> 1] >> (16 * (j & 1))) & 0xffffu);
      }
    }
  }
}

// lane ^ MASK exchange inside groups of 16 lanes without an LDS address: DPP quad_perm (1, 2), row_ror:8 (8), ds_swizzle (4)
template <int MASK>
__device__ __forceinline__ float lane_xor16(float v) {
  int x = __builtin_bit_cast(int, v);
  if constexpr (MASK == 1) x = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);
  else if constexpr (MASK == 2) x = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);
  else if constexpr (MASK == 8) x = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true);
  else x = __builtin_amdgcn_ds_swizzle(x, (MASK << 10) | 0x1F);
  return __builtin_bit_cast(float, x);
}
// CH lanes (c = lane % CH) each hold CH partial sums part[i]; on return part[0] is the FULL sum of value i == c.
// Recursive halving: CH - 1 exchanges instead of CH * log2(CH) for one butterfly per value.
template <int CH>
__device__ __forceinline__ float reduce_scatter_lanes(float (&part)[CH], int c) {
#define VT_RS_STEP(N)                                                   \
  if constexpr (CH >= 2 * (N)) {                                        \
    const bool up = (c & (N)) != 0;                                     \
    _Pragma("unroll") for (int j = 0; j < (N); ++j) {                   \
      const float send = up ? part[j] : part[j + (N)];                  \
      const float mine = up ? part[j + (N)] : part[j];                  \
      part[j] = mine + lane_xor16<(N)>(send);                           \
    }                                                                   \
  }
  VT_RS_STEP(8)
  VT_RS_STEP(4)
  VT_RS_STEP(2)
  VT_RS_STEP(1)
#undef VT_RS_STEP
  return part[0];
}
template <int CH>
__device__ __forceinline__ float allreduce_lanes(float v) {
  if constexpr (CH >= 16) v += lane_xor16<8>(v);
  if constexpr (CH >= 8) v += lane_xor16<4>(v);
  v += lane_xor16<2>(v);
  v += lane_xor16<1>(v);
  return v;
}

// ------------------------------------------------------------------------------------------------------------------
// attn_decode_kernel: single-query attention over the paged tiles, flash-decoding style. HBM-bound: the whole job is
// to stream K and V^T tiles once, fully coalesced, with enough waves in flight.
//   grid (head, sequence, split); 4 waves per block; wave w of split s owns tiles t = 4*s + w, 4*s + w + 4*nsplit, ...
//   scores : 16 (HD=128) lanes cooperate on one 256-B K row, 4 rows per wave-instruction; the 16 partial sums per lane are
//            reduce-scattered over the 16 lanes (15 DPP/swizzle exchanges per tile), lane ends up owning key (lane%16)*4 + lane/16
//   PV     : 8 lanes cooperate on one 128-B V^T row (8 rows per wave-instruction = 8 full cache lines); every lane
//            keeps HD/8 partial accumulators and the cross-lane reduction happens ONCE after the last tile
//   each block writes an online-softmax partial (m, l, o[HD]) to scratch; attn_decode_combine_kernel merges the splits.
// ------------------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ Q, int ldq,
                                                          const bf16_t* __restrict__ Kt,
                                                          const bf16_t* __restrict__ Vt,
                                                          const int* __restrict__ tile_table,
                                                          const VtAttnSeq* __restrict__ seqs, int heads,
                                                          float scale_log2e, float* __restrict__ part, int nsplit) {
  constexpr int CH = HD / 8;        // 16-B chunks per K row; CH lanes cooperate on one key
  constexpr int KPI = 64 / CH;      // keys per wave-instruction
  constexpr int NACC = HD / 8;      // V^T rows handled per lane (one per PV instruction)
  __shared__ float sm_m[4], sm_l[4];
  __shared__ float sm_o[4][HD];
  __shared__ __attribute__((aligned(16))) float sm_p[4][64];
  const VtAttnSeq sq = seqs[blockIdx.y];
  const int head = blockIdx.x, split = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = (sq.kv_len + 63) >> 6;
  const bf16_t* qp = Q + (size_t)sq.q_row0 * ldq + head * HD;
  const int c = lane % CH;
  const u32x4 qv = *(const u32x4*)(qp + c * 8);
  float qf[8];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    qf[2 * w] = oplo_to_f32(qv[w]);
    qf[2 * w + 1] = ophi_to_f32(qv[w]);
  }
  float m_run = -INFINITY, l_run = 0.f, acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
  const int vrow = lane >> 3, vchk = lane & 7;
  const int key_of_lane = c * KPI + lane / CH;   // the key (inside a tile) whose score this lane ends up owning

  for (int t = split * 4 + wave; t < ntiles; t += 4 * nsplit) {
    const size_t toff = ((size_t)tile_table[sq.table_off + t] * heads + head) * 64 * HD;
    const bf16_t* kt = Kt + toff;
    const bf16_t* vt = Vt + toff;
    // the whole tile (16 KiB of K + 16 KiB of V^T) is requested before anything is consumed
    u32x4 kk[CH], vv[NACC];
#pragma unroll
    for (int i = 0; i < CH; ++i) kk[i] = __builtin_nontemporal_load((const u32x4*)(kt + (i * KPI + lane / CH) * HD + c * 8));
#pragma unroll
    for (int i = 0; i < NACC; ++i) vv[i] = __builtin_nontemporal_load((const u32x4*)(vt + (i * 8 + vrow) * 64 + vchk * 8));
    __builtin_amdgcn_sched_barrier(0);
    float part[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const u32x4 kv = kk[i];
      float part_s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        part_s = fmaf(oplo_to_f32(kv[w]), qf[2 * w], part_s);
        part_s = fmaf(ophi_to_f32(kv[w]), qf[2 * w + 1], part_s);
      }
      part[i] = part_s;
    }
    const float s_mine = reduce_scatter_lanes<CH>(part, c);   // full score of key c*KPI + lane/CH
    const int mykey = t * 64 + key_of_lane;
    const float s2 = (mykey < sq.kv_len) ? s_mine * scale_log2e : -INFINITY;
    const float m_new = fmaxf(m_run, wave_max(s2));
    const float alpha = fast_exp2(m_run - m_new);
    const float p = fast_exp2(s2 - m_new);
    l_run = l_run * alpha + wave_sum(p);
    m_run = m_new;
    sm_p[wave][key_of_lane] = p;
    __builtin_amdgcn_wave_barrier();  // DS ops of one wave are in order; only stop compiler reordering
    const f32x4 pa = *(const f32x4*)(&sm_p[wave][vchk * 8]);
    const f32x4 pb = *(const f32x4*)(&sm_p[wave][vchk * 8 + 4]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      float a = acc[i] * alpha;
      a = fmaf(f16lo_to_f32(vv[i][0]), pa[0], a);
      a = fmaf(f16hi_to_f32(vv[i][0]), pa[1], a);
      a = fmaf(f16lo_to_f32(vv[i][1]), pa[2], a);
      a = fmaf(f16hi_to_f32(vv[i][1]), pa[3], a);
      a = fmaf(f16lo_to_f32(vv[i][2]), pb[0], a);
      a = fmaf(f16hi_to_f32(vv[i][2]), pb[1], a);
      a = fmaf(f16lo_to_f32(vv[i][3]), pb[2], a);
      a = fmaf(f16hi_to_f32(vv[i][3]), pb[3], a);
      acc[i] = a;
    }
  }
  // reduce the 8 lanes of every V^T row, then combine the 4 waves through LDS
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    float a = acc[i];
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    a += __shfl_xor(a, 4, 64);
    if (vchk == 0) sm_o[wave][i * 8 + vrow] = a;
  }
  if (lane == 0) {
    sm_m[wave] = m_run;
    sm_l[wave] = l_run;
  }
  __syncthreads();
  if (wave == 0) {
    const float m = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
    float l = 0.f, w4[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      w4[w] = (sm_m[w] == -INFINITY) ? 0.f : fast_exp2(sm_m[w] - m);
      l += sm_l[w] * w4[w];
    }
    float* dst = part + (((size_t)blockIdx.y * heads + head) * nsplit + split) * (HD + 2);
    if (lane == 0) {
      dst[0] = m;
      dst[1] = l;
    }
#pragma unroll
    for (int i = 0; i < HD / 64; ++i) {
      const int d = lane + 64 * i;
      float o = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) o += sm_o[w][d] * w4[w];
      dst[2 + d] = o;
    }
  }
}

template <int HD>
__global__ __launch_bounds__(64) void attn_decode_combine_kernel(const float* __restrict__ part,
                                                                 const VtAttnSeq* __restrict__ seqs,
                                                                 bf16_t* __restrict__ O, int ldo, int heads, int nsplit) {
  const int head = blockIdx.x, seq = blockIdx.y, lane = threadIdx.x;
  const float* src = part + ((size_t)seq * heads + head) * nsplit * (HD + 2);
  float m = -INFINITY;
  for (int s = 0; s < nsplit; ++s) m = fmaxf(m, src[s * (HD + 2)]);
  float l = 0.f, o[HD / 64];
#pragma unroll
  for (int i = 0; i < HD / 64; ++i) o[i] = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float ms = src[s * (HD + 2)];
    const float w = (ms == -INFINITY) ? 0.f : fast_exp2(ms - m);
    l += src[s * (HD + 2) + 1] * w;
#pragma unroll
    for (int i = 0; i < HD / 64; ++i) o[i] += src[s * (HD + 2) + 2 + lane + 64 * i] * w;
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
  bf16_t* op = O + (size_t)seqs[seq].q_row0 * ldo + head * HD;
#pragma unroll
  for (int i = 0; i < HD / 64; ++i) op[lane + 64 * i] = f32_to_op(o[i] * inv);
}

// ------------------------------------------------------------------------------------------------------------------
// attn_decode_fused_kernel: the whole attention part of one decode step of one layer in ONE launch (q_len == 1):
//   rotary embedding of the new q and k rows, append of the new k row / v column to the paged tiles, single-query
//   attention over the cache, combine. Replaces kv_tiles + attn_decode + attn_decode_combine (3 launches, 2 boundaries).
//   grid (head, sequence), 8 waves per block: wave w owns tiles w, w + 8, ... of its (sequence, head) and streams them
//   exactly like attn_decode_kernel (K rows by 16-lane groups, V^T rows by 8-lane groups, everything of a tile in
//   flight at once); the 8 online-softmax partials meet in LDS -- no scratch, no cross-block traffic. One head of one
//   sequence is <= 1 MiB of K/V^T at 2048 tokens, which one CU streams in a few microseconds.
//   The wave that owns the LAST tile also owns the new token: it rotates k, scores it from registers, stores the k row
//   and the v column, and adds p_new * v_new to its accumulators. A tile that STARTS with the new token is zero-filled
//   around it (same invariant as kv_tiles_kernel: padding rows/columns of a tile are zero, never garbage).
// ------------------------------------------------------------------------------------------------------------------
template <int HD, bool ROPE>
__global__ __launch_bounds__(512) void attn_decode_fused_kernel(
    const bf16_t* __restrict__ qkv, int ldqkv, int q_col0, int k_col0, int v_col0, bf16_t* __restrict__ Kt,
    bf16_t* __restrict__ Vt, const int* __restrict__ tile_table, const VtAttnSeq* __restrict__ seqs, int heads,
    const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, const int* __restrict__ positions,
    float scale_log2e, bf16_t* __restrict__ O, int ldo) {
  constexpr int NW = 8;
  constexpr int CH = HD / 8;        // 16-B chunks per K row; CH lanes cooperate on one key
  constexpr int KPI = 64 / CH;      // keys per wave-instruction
  constexpr int NACC = HD / 8;      // V^T rows handled per lane
  __shared__ float sm_m[NW], sm_l[NW];
  __shared__ float sm_o[NW][HD];
  __shared__ __attribute__((aligned(16))) float sm_p[NW][64];
  __shared__ __attribute__((aligned(16))) bf16_t sm_v[HD];
  const VtAttnSeq sq = seqs[blockIdx.y];
  const int head = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int past = sq.kv_len - 1;
  const int t_last = past >> 6, r_new = past & 63, ntiles = t_last + 1;
  const bf16_t* qrow = qkv + (size_t)sq.q_row0 * ldqkv;
  const int c = lane % CH, ch = c % (CH / 2);
  const bool upper = c >= CH / 2;
  constexpr bool rope = ROPE;
  const int vrow = lane >> 3, vchk = lane & 7;
  const int key_of_lane = c * KPI + lane / CH;   // the key (inside a tile) whose score this lane ends up owning
  float m_run = -INFINITY, l_run = 0.f, acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;

  // the whole tile (16 KiB of K + 16 KiB of V^T) is requested before anything is consumed; the first tile's requests go
  // out before the rotary prologue so that its three dependent look-ups (positions -> tables, q row) overlap the K/V latency
  u32x4 kk[CH], vv[NACC];
  auto issue = [&](int t) {
    const size_t toff = ((size_t)tile_table[sq.table_off + t] * heads + head) * 64 * HD;
    if (t == t_last && r_new == 0) {   // tile starts with the new token: nothing to read (it is zero-filled below)
#pragma unroll
      for (int i = 0; i < CH; ++i) kk[i] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < NACC; ++i) vv[i] = (u32x4){0u, 0u, 0u, 0u};
    } else {
#pragma unroll
      for (int i = 0; i < CH; ++i)
        kk[i] = __builtin_nontemporal_load((const u32x4*)(Kt + toff + (i * KPI + lane / CH) * HD + c * 8));
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        vv[i] = __builtin_nontemporal_load((const u32x4*)(Vt + toff + (i * 8 + vrow) * 64 + vchk * 8));
    }
    return toff;
  };
  float cs[8], sn[8];
  if (rope) {
    const int rp = positions[sq.q_row0];
    const float* cp = rope_cos + (size_t)rp * (HD / 2) + ch * 8;
    const float* sp = rope_sin + (size_t)rp * (HD / 2) + ch * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cs[j] = cp[j];
      sn[j] = sp[j];
    }
  }
  // this lane's 16-B chunk c of a head row, rotated (half-split rotary, same arithmetic and bf16 rounding as kv_tiles_kernel)
  auto rotate = [&](const u32x4 lo, const u32x4 hi) -> u32x4 {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float a0 = oplo_to_f32(lo[w]), a1 = ophi_to_f32(lo[w]);
      const float b0 = oplo_to_f32(hi[w]), b1 = ophi_to_f32(hi[w]);
      const float c0 = cs[2 * w], c1 = cs[2 * w + 1], s0 = sn[2 * w], s1 = sn[2 * w + 1];
      o[w] = upper ? pack_op2(rope_hi(a0, b0, c0, s0), rope_hi(a1, b1, c1, s1)) : pack_op2(rope_lo(a0, b0, c0, s0), rope_lo(a1, b1, c1, s1));
    }
    return o;
  };
  auto head_chunk = [&](const bf16_t* base) -> u32x4 {
    if (!rope) return *(const u32x4*)(base + c * 8);
    return rotate(*(const u32x4*)(base + ch * 8), *(const u32x4*)(base + HD / 2 + ch * 8));
  };
  // raw q chunk(s) first, then the first tile's 32 requests, then the rotary arithmetic: loads complete in issue order, so
  // the prologue only ever waits for its own small look-ups while the tile is in flight
  const bf16_t* qbase = qrow + q_col0 + head * HD;
  const u32x4 q_lo = *(const u32x4*)(qbase + (rope ? ch : c) * 8);
  const u32x4 q_hi = *(const u32x4*)(qbase + (rope ? HD / 2 + ch * 8 : c * 8));
  __builtin_amdgcn_sched_barrier(0);
  int t = wave;
  size_t toff = 0;
  if (t < ntiles) toff = issue(t);
  __builtin_amdgcn_sched_barrier(0);
  const u32x4 qv = rope ? rotate(q_lo, q_hi) : q_lo;
  float qf[8];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    qf[2 * w] = oplo_to_f32(qv[w]);
    qf[2 * w + 1] = ophi_to_f32(qv[w]);
  }
  __builtin_amdgcn_sched_barrier(0);

  while (t < ntiles) {
    const bool is_last = t == t_last;
    const bool fresh = is_last && r_new == 0;
    bf16_t* kt = Kt + toff;
    bf16_t* vt = Vt + toff;
    float part[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const u32x4 kv = kk[i];
      float part_s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        part_s = fmaf(oplo_to_f32(kv[w]), qf[2 * w], part_s);
        part_s = fmaf(ophi_to_f32(kv[w]), qf[2 * w + 1], part_s);
      }
      part[i] = part_s;
    }
    float s_mine = reduce_scatter_lanes<CH>(part, c);   // full score of key_of_lane = c*KPI + lane/CH
    if (is_last) {
      // ---- the new token: key r_new of this tile ----
      const u32x4 kr = head_chunk(qrow + k_col0 + head * HD);
      float part_s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        part_s = fmaf(oplo_to_f32(kr[w]), qf[2 * w], part_s);
        part_s = fmaf(ophi_to_f32(kr[w]), qf[2 * w + 1], part_s);
      }
      part_s = allreduce_lanes<CH>(part_s);
      if (key_of_lane == r_new) s_mine = part_s;
      if (lane < CH) {
        *(u32x4*)(kt + r_new * HD + lane * 8) = kr;                                       // lane < CH: c == lane
        const u32x4 vn = *(const u32x4*)(qrow + v_col0 + head * HD + lane * 8);       // bf16 from the projection -> fp16 page format
        *(u32x4*)(&sm_v[lane * 8]) = (u32x4){op2_to_f16x2(vn.x), op2_to_f16x2(vn.y), op2_to_f16x2(vn.z), op2_to_f16x2(vn.w)};
      }
      __builtin_amdgcn_wave_barrier();
      if (!fresh) {
#pragma unroll
        for (int i = 0; i < HD / 64; ++i) vt[(lane + 64 * i) * 64 + r_new] = sm_v[lane + 64 * i];
      } else {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int it = lane; it < 63 * CH; it += 64) *(u32x4*)(kt + HD + it * 8) = z;      // K rows 1..63
        for (int it = lane; it < HD * 8; it += 64) {                                       // V^T: (d, 8-key chunk)
          const int d = it >> 3, kc = it & 7;
          u32x4 w = z;
          if (kc == 0) w.x = sm_v[d];
          *(u32x4*)(vt + d * 64 + kc * 8) = w;
        }
      }
    }
    const int mykey = t * 64 + key_of_lane;
    const float s2 = (mykey < sq.kv_len) ? s_mine * scale_log2e : -INFINITY;
    const float m_new = fmaxf(m_run, wave_max(s2));
    const float alpha = fast_exp2(m_run - m_new);
    const float p = fast_exp2(s2 - m_new);
    l_run = l_run * alpha + wave_sum(p);
    m_run = m_new;
    sm_p[wave][key_of_lane] = p;
    __builtin_amdgcn_wave_barrier();  // DS ops of one wave are in order; only stop compiler reordering
    float p_new = 0.f;
    if (is_last) {
      // the new token's value comes from sm_v below, never from the loaded tile: its weight is taken out of the tile's row of
      // probabilities, so a column r_new that already holds a value (a step that was rolled back and is run again) is not
      // counted twice and the kernel is idempotent
      p_new = sm_p[wave][r_new];
      __builtin_amdgcn_wave_barrier();
      if (key_of_lane == r_new) sm_p[wave][r_new] = 0.f;
      __builtin_amdgcn_wave_barrier();
    }
    const f32x4 pa = *(const f32x4*)(&sm_p[wave][vchk * 8]);
    const f32x4 pb = *(const f32x4*)(&sm_p[wave][vchk * 8 + 4]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      float a = acc[i] * alpha;
      a = fmaf(f16lo_to_f32(vv[i][0]), pa[0], a);
      a = fmaf(f16hi_to_f32(vv[i][0]), pa[1], a);
      a = fmaf(f16lo_to_f32(vv[i][1]), pa[2], a);
      a = fmaf(f16hi_to_f32(vv[i][1]), pa[3], a);
      a = fmaf(f16lo_to_f32(vv[i][2]), pb[0], a);
      a = fmaf(f16hi_to_f32(vv[i][2]), pb[1], a);
      a = fmaf(f16lo_to_f32(vv[i][3]), pb[2], a);
      a = fmaf(f16hi_to_f32(vv[i][3]), pb[3], a);
      acc[i] = a;
    }
    if (is_last && vchk == 0) {   // column r_new of the loaded tile carries no weight (see above): add the new value here
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = fmaf(f16_bits_to_f32(sm_v[i * 8 + vrow]), p_new, acc[i]);
    }
    t += NW;
    if (t < ntiles) toff = issue(t);
  }
  // reduce the 8 lanes of every V^T row, then combine the waves through LDS
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    float a = acc[i];
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    a += __shfl_xor(a, 4, 64);
    if (vchk == 0) sm_o[wave][i * 8 + vrow] = a;
  }
  if (lane == 0) {
    sm_m[wave] = m_run;
    sm_l[wave] = l_run;
  }
  __syncthreads();
  if (threadIdx.x < HD) {
    float m = sm_m[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, sm_m[w]);
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float wt = (sm_m[w] == -INFINITY) ? 0.f : fast_exp2(sm_m[w] - m);
      l += sm_l[w] * wt;
      o += sm_o[w][threadIdx.x] * wt;
    }
    O[(size_t)sq.q_row0 * ldo + head * HD + threadIdx.x] = f32_to_op(l > 0.f ? o / l : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// attn_temporal_kernel: qkv bf16 [B*T*N][3D], row (b*T+t)*N+n; one wave per (b, n, head), lane = d (hd = 64).
// q is pre-scaled by hd^-0.5 in the packed weights.
// ------------------------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(256) void attn_temporal_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            int B, int N, int heads) {
  const int D = heads * 64;
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long total = (long)B * N * heads;
  if (wid >= total) return;
  const int head = (int)(wid % heads);
  const long bn = wid / heads;
  const int n = (int)(bn % N), b = (int)(bn / N);
  float q[T], k[T], v[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const size_t row = ((size_t)b * T + t) * N + n;
    const bf16_t* p = qkv + row * (size_t)(3 * D) + head * 64 + lane;
    q[t] = op_to_f32(p[0]);
    k[t] = op_to_f32(p[D]);
    v[t] = op_to_f32(p[2 * D]);
  }
#pragma unroll
  for (int t1 = 0; t1 < T; ++t1) {
    float s[T];
    float mx = -INFINITY;
#pragma unroll
    for (int t2 = 0; t2 < T; ++t2) {
      s[t2] = wave_sum(q[t1] * k[t2]);
      mx = fmaxf(mx, s[t2]);
    }
    float den = 0.f, acc = 0.f;
#pragma unroll
    for (int t2 = 0; t2 < T; ++t2) {
      const float p = __expf(s[t2] - mx);
      den += p;
      acc += p * v[t2];
    }
    const size_t row = ((size_t)b * T + t1) * N + n;
    out[row * (size_t)D + head * 64 + lane] = f32_to_op(acc / den);
  }
}

// precise level 2 (vt_vit_model.precise): the same attention on the fp32 q | k | v of the pair projection, everything in fp32, the output
// as an operand pair (out_proj's A operand); one lane per head dim like the generic kernel (a verification mode: speed is not the point)
template <int T>
__global__ __launch_bounds__(256) void attn_temporal_f32_kernel(const float* __restrict__ qkv, int ld, bf16_t* __restrict__ out,
                                                                bf16_t* __restrict__ out_lo, int B, int N, int heads) {
  const int D = heads * 64;
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long total = (long)B * N * heads;
  if (wid >= total) return;
  const int head = (int)(wid % heads);
  const long bn = wid / heads;
  const int n = (int)(bn % N), b = (int)(bn / N);
  float q[T], k[T], v[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const size_t row = ((size_t)b * T + t) * N + n;
    const float* p = qkv + row * (size_t)ld + head * 64 + lane;
    q[t] = p[0];
    k[t] = p[D];
    v[t] = p[2 * D];
  }
#pragma unroll
  for (int t1 = 0; t1 < T; ++t1) {
    float s[T];
    float mx = -INFINITY;
#pragma unroll
    for (int t2 = 0; t2 < T; ++t2) {
      s[t2] = wave_sum(q[t1] * k[t2]);
      mx = fmaxf(mx, s[t2]);
    }
    float den = 0.f, acc = 0.f;
#pragma unroll
    for (int t2 = 0; t2 < T; ++t2) {
      const float p = __expf(s[t2] - mx);
      den += p;
      acc += p * v[t2];
    }
    const size_t row = ((size_t)b * T + t1) * N + n;
    const float o = acc / den;
    const bf16_t hi = f32_to_op(o);
    out[row * (size_t)D + head * 64 + lane] = hi;
    out_lo[row * (size_t)D + head * 64 + lane] = f32_to_op(o - op_to_f32(hi));
  }
}

// T == 8 (the LanguageBind video tower): one wave per (clip, position, head) with 16-byte accesses only.
//   load  : lane (t = lane>>3, c = lane&7) fetches the 16-B chunk c of frame t's q, k and v head rows (8 lanes = one 128-B row)
//           into a per-wave LDS image (rows padded to 144 B: conflict-free for the reads below),
//   scores: lane (t1 = lane>>3, t2 = lane&7) owns S[t1][t2] = q[t1] . k[t2] (8 x two ds_read_b128), softmax over the 8 lanes of
//           a row with three xor-shuffles, P stays fp32,
//   output: lane (t1, c) owns out[t1][8c .. 8c+8) = sum_t2 P[t1][t2] v[t2][8c ..]; one 16-B store per lane.
// The generic kernel above (one lane per head dim, 2-byte accesses, 64 wave reductions per item) took 45 us per layer on the
// 8 x 576 x 16 items of a 336 px clip; this one is bandwidth-bound.
__global__ __launch_bounds__(256) void attn_temporal8_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int B, int N,
                                                             int heads) {
  constexpr int T = 8, RS = 72;                       // row stride in bf16 elements (144 B)
  __shared__ __attribute__((aligned(16))) bf16_t sm[4][3][T][RS];
  __shared__ __attribute__((aligned(16))) float sp[4][T][T];
  const int D = heads * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wid = (long)blockIdx.x * 4 + wave;
  const long total = (long)B * N * heads;
  if (wid >= total) return;
  const int head = (int)(wid % heads);
  const long bn = wid / heads;
  const int n = (int)(bn % N), b = (int)(bn / N);
  const int t = lane >> 3, c = lane & 7;
  const size_t row = ((size_t)b * T + t) * N + n;
  const bf16_t* src = qkv + row * (size_t)(3 * D) + head * 64 + c * 8;
  const u32x4 qv = *(const u32x4*)src, kv = *(const u32x4*)(src + D), vv = *(const u32x4*)(src + 2 * D);
  *(u32x4*)&sm[wave][0][t][c * 8] = qv;
  *(u32x4*)&sm[wave][1][t][c * 8] = kv;
  *(u32x4*)&sm[wave][2][t][c * 8] = vv;
  __builtin_amdgcn_wave_barrier();                    // DS operations of one wave are ordered; only the compiler must not reorder
  // ---- S[t1][t2], t1 = lane>>3, t2 = lane&7 ----
  float sc = 0.f;
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    const u32x4 a = *(const u32x4*)&sm[wave][0][t][ch * 8];
    const u32x4 k8 = *(const u32x4*)&sm[wave][1][c][ch * 8];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      sc = fmaf(oplo_to_f32(a[w]), oplo_to_f32(k8[w]), sc);
      sc = fmaf(ophi_to_f32(a[w]), ophi_to_f32(k8[w]), sc);
    }
  }
  float mx = sc;
  mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
  const float pe = __expf(sc - mx);
  float den = pe;
  den += __shfl_xor(den, 1, 64);
  den += __shfl_xor(den, 2, 64);
  den += __shfl_xor(den, 4, 64);
  sp[wave][t][c] = pe / den;
  __builtin_amdgcn_wave_barrier();
  // ---- out[t1][8c .. 8c+8) ----
  const f32x4 p0 = *(const f32x4*)&sp[wave][t][0], p1 = *(const f32x4*)&sp[wave][t][4];
  const float pr[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
  for (int t2 = 0; t2 < T; ++t2) {
    const u32x4 v8 = *(const u32x4*)&sm[wave][2][t2][c * 8];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      acc[2 * w] = fmaf(pr[t2], oplo_to_f32(v8[w]), acc[2 * w]);
      acc[2 * w + 1] = fmaf(pr[t2], ophi_to_f32(v8[w]), acc[2 * w + 1]);
    }
  }
  u32x4 o;
  o.x = pack_op2(acc[0], acc[1]);
  o.y = pack_op2(acc[2], acc[3]);
  o.z = pack_op2(acc[4], acc[5]);
  o.w = pack_op2(acc[6], acc[7]);
  *(u32x4*)(out + row * (size_t)D + head * 64 + c * 8) = o;
}

}  // namespace

int g_vt_flash_attn_kernel = 0;
int g_vt_flash_attn_wgs = 0;
int g_vt_flash_attn_order = 1;

int vt_flash_attn_launch(const bf16_t* Q, int ldq, const bf16_t* Kt, const bf16_t* Vt, const int* tile_table,
                         const VtAttnSeq* seqs, int nseq, int max_q_len, bf16_t* O, int ldo, int heads, int HD,
                         int causal, float scale, hipStream_t s, uint8_t* O4, uint8_t* oexp) {
  VT_REQUIRE(Q && Kt && Vt && tile_table && seqs && O, "vt_flash_attn: null pointer");
  VT_REQUIRE(!O4 || (oexp && HD == 128 && causal && ldo == heads * HD && (((size_t)O4) & 7) == 0),
             "vt_flash_attn: the level 3 operand out needs head_dim 128, a causal launch and a dense output (ldo == heads * 128)");
  VT_REQUIRE(HD == 64 || HD == 128, "vt_flash_attn: head_dim %d unsupported (64 or 128)", HD);
  VT_REQUIRE(nseq > 0 && max_q_len > 0 && heads > 0, "vt_flash_attn: empty problem");
  VT_REQUIRE(ldq % 8 == 0 && ldo % 4 == 0, "vt_flash_attn: ldq %% 8 and ldo %% 4 must be 0");
  const float sl2 = scale * 1.4426950408889634f;
  // algorithmic FLOP are not known here without reading seq_desc back; the caller-side bench computes them.
  VtProfScope prof(VT_PROF_FLASH_ATTN, 0.0, s);
  // head_dim 128, long sequences: the one-wave-per-SIMD kernel on 256-row blocks (vt_attn_w4.hip) once its grid fills the chip --
  // (heads x sequences x 256-row blocks) >= 256 workgroups, one per CU; shorter prompts keep the 128-row blocks below (two per CU)
  if (HD == 128 && g_vt_flash_attn_kernel != 1) {
    const bool aligned = ldo % 8 == 0 && (((size_t)O) & 15) == 0;   // it stores 16-byte chunks
    const bool fills = (long)heads * nseq * cdiv(max_q_len, 256) >= 256 && max_q_len >= 1024;
    if (g_vt_flash_attn_kernel >= 2 || (fills && aligned)) {
      VT_REQUIRE(aligned, "vt_flash_attn: the one-wave-per-SIMD kernel stores 16-byte chunks: O 16-byte aligned, ldo %% 8 == 0");
      return vt_flash_attn_w4_launch(Q, ldq, Kt, Vt, tile_table, seqs, nseq, max_q_len, O, ldo, heads, causal, sl2,
                                     g_vt_flash_attn_kernel == 3 ? 0 : g_vt_flash_attn_kernel == 4 ? 2 : 1, s, O4, oexp);
    }
  }
  // long sequences: 256-row blocks, 3-stage ring; short ones (ViT frames, small prefills): 128-row blocks
  // measured (tools/attn_bench.py, S=5120 causal): 128-row blocks (2 per CU) 272 us vs 256-row blocks 288 us -- the finer
  // causal granularity wins; the 8-wave / 3-stage variant stays selectable for experiments
#ifdef VT_ABLATIONS   // experiment switches exist only in the test library (python -m vitron_amd.build --ablations)
  static const int use_big = getenv("VT_FLASH_QBLK256") ? atoi(getenv("VT_FLASH_QBLK256")) : 0;
  const bool big = max_q_len >= 1024 && use_big;
#else
  (void)0;
#endif
#define VT_FA(HDV, CV, NW, NS, ...)                                                                            \
  do {                                                                                                         \
    auto kern = flash_attn_kernel<HDV, CV, NW, NS __VA_OPT__(,) __VA_ARGS__>;                                  \
    const int smem = NS * 2 * 64 * HDV * 2;                                                                    \
    VT_LDS_ATTR_ONCE(kern, smem);                                                                              \
    dim3 grid(heads, cdiv(max_q_len, 32 * NW), nseq), block(64 * NW);                                          \
    hipLaunchKernelGGL(kern, grid, block, smem, s, Q, ldq, Kt, Vt, tile_table, seqs, O, ldo, heads, sl2, O4, oexp); \
  } while (0)
  if (O4) {
    VT_FA(128, true, 4, 2, 0, true);
  } else if (HD == 64) {
    if (causal) VT_FA(64, true, 4, 2); else VT_FA(64, false, 4, 2);
#ifdef VT_ABLATIONS
  } else if (big) {
    if (causal) VT_FA(128, true, 8, 3); else VT_FA(128, false, 8, 3);
#endif
  } else {
#ifdef VT_ABLATIONS
    static const int abl = getenv("VT_FLASH_ABL") ? atoi(getenv("VT_FLASH_ABL")) : 0;   // timing ablations: VT_FLASH_ABL=1|2|3 python tools/attn_bench.py
    if (causal && abl == 1) VT_FA(128, true, 4, 2, 1);
    else if (causal && abl == 2) VT_FA(128, true, 4, 2, 2);
    else if (causal && abl == 3) VT_FA(128, true, 4, 2, 3);
    else if (causal && abl == 4) VT_FA(128, true, 4, 2, 4);
    else if (causal && abl == 8) VT_FA(128, true, 4, 2, 8);
    else
#endif
    if (causal) VT_FA(128, true, 4, 2); else VT_FA(128, false, 4, 2);
  }
#undef VT_FA
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_kv_tiles_launch(bf16_t* qkv, int ldqkv, int q_col0, int k_col0, int v_col0, bf16_t* Kt, bf16_t* Vt,
                       const int* tile_table, const VtAttnSeq* seqs, int nseq, int max_new_tiles, int heads, int HD,
                       const float* rope_cos, const float* rope_sin, const int* positions, hipStream_t s) {
  VT_REQUIRE(qkv && Kt && Vt && tile_table && seqs, "vt_kv_tiles: null pointer");
  VT_REQUIRE(HD == 64 || HD == 128, "vt_kv_tiles: head_dim %d unsupported", HD);
  VT_REQUIRE(ldqkv % 8 == 0 && q_col0 % 8 == 0 && k_col0 % 8 == 0 && v_col0 % 8 == 0, "vt_kv_tiles: misaligned columns");
  dim3 grid(max_new_tiles, heads, nseq), block(256);
  const bool rope = rope_cos != nullptr;
  if (rope) VT_REQUIRE(rope_sin && positions, "vt_kv_tiles: rope needs sin table and positions");
  if (HD == 64) {
    if (rope) hipLaunchKernelGGL((kv_tiles_kernel<64, true>), grid, block, 0, s, qkv, ldqkv, q_col0, k_col0, v_col0, Kt, Vt, tile_table, seqs, heads, rope_cos, rope_sin, positions);
    else hipLaunchKernelGGL((kv_tiles_kernel<64, false>), grid, block, 0, s, qkv, ldqkv, q_col0, k_col0, v_col0, Kt, Vt, tile_table, seqs, heads, rope_cos, rope_sin, positions);
  } else {
    if (rope) hipLaunchKernelGGL((kv_tiles_kernel<128, true>), grid, block, 0, s, qkv, ldqkv, q_col0, k_col0, v_col0, Kt, Vt, tile_table, seqs, heads, rope_cos, rope_sin, positions);
    else hipLaunchKernelGGL((kv_tiles_kernel<128, false>), grid, block, 0, s, qkv, ldqkv, q_col0, k_col0, v_col0, Kt, Vt, tile_table, seqs, heads, rope_cos, rope_sin, positions);
  }
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_kv_tiles_precise_launch(const float* qk32, int ld32, int v_f32, bf16_t* qkv, int ldqkv, int q_col0, int v_col0, bf16_t* qlo, bf16_t* Kt, bf16_t* Vt,
                               bf16_t* klo, const int* tile_table, const VtAttnSeq* seqs, int nseq, int max_new_tiles, int heads, int HD,
                               const float* rope_cos, const float* rope_sin, const int* positions, hipStream_t s) {
  VT_REQUIRE(qk32 && qkv && qlo && Kt && Vt && klo && tile_table && seqs, "vt_kv_tiles_precise: null pointer");
  if (rope_cos) VT_REQUIRE(rope_sin && positions, "vt_kv_tiles_precise: rope needs sin table and positions");
  VT_REQUIRE(HD == 128 || HD == 64, "vt_kv_tiles_precise: head_dim %d unsupported (64, 128)", HD);
  VT_REQUIRE(ldqkv % 8 == 0 && q_col0 % 8 == 0 && v_col0 % 8 == 0, "vt_kv_tiles_precise: misaligned columns");
  dim3 grid(max_new_tiles, heads, nseq), block(256);
  VT_REQUIRE(ld32 >= (v_f32 ? 3 : 2) * heads * HD && ld32 % 4 == 0, "vt_kv_tiles_precise: ld32 %d too small", ld32);
  if (HD == 128)
    hipLaunchKernelGGL((kv_tiles_precise_kernel<128>), grid, block, 0, s, qk32, ld32, v_f32, qkv, ldqkv, q_col0, v_col0, qlo, Kt, Vt, klo, tile_table, seqs,
                       heads, rope_cos, rope_sin, positions);
  else
    hipLaunchKernelGGL((kv_tiles_precise_kernel<64>), grid, block, 0, s, qk32, ld32, v_f32, qkv, ldqkv, q_col0, v_col0, qlo, Kt, Vt, klo, tile_table, seqs,
                       heads, rope_cos, rope_sin, positions);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_flash_attn_precise_launch(const bf16_t* Q, int ldq, const bf16_t* Qlo, int ldqlo, const bf16_t* Kt, const bf16_t* Klo,
                                 int klo_tiles_per_seq, const bf16_t* Vt, const int* tile_table, const VtAttnSeq* seqs, int nseq,
                                 int max_q_len, bf16_t* O, int ldo, bf16_t* Olo, int heads, int HD, int causal, float scale, hipStream_t s) {
  VT_REQUIRE(Q && Qlo && Kt && Klo && Vt && tile_table && seqs && O, "vt_flash_attn_precise: null pointer");
  VT_REQUIRE(HD == 128 || HD == 64, "vt_flash_attn_precise: head_dim %d unsupported (64, 128)", HD);
  VT_REQUIRE(nseq > 0 && max_q_len > 0 && heads > 0 && klo_tiles_per_seq > 0, "vt_flash_attn_precise: empty problem");
  VT_REQUIRE(ldq % 8 == 0 && ldqlo % 8 == 0 && ldo % 4 == 0, "vt_flash_attn_precise: ldq %% 8, ldqlo %% 8 and ldo %% 4 must be 0");
  const float sl2 = scale * 1.4426950408889634f;
  VtProfScope prof(VT_PROF_FLASH_ATTN, 0.0, s);
  const int smem = 2 * 3 * 64 * HD * 2;
  int dev = 0;
  VT_HIP(hipGetDevice(&dev));
  VT_REQUIRE(dev >= 0 && dev < 64, "vt_flash_attn_precise: device ordinal %d", dev);
  dim3 grid(heads, cdiv(max_q_len, 128), nseq), block(256);
#define VT_FAP(CV, HDV)                                                                                        \
  do {                                                                                                         \
    auto kern = flash_attn_precise_kernel<CV, HDV>;                                                            \
    static std::atomic<unsigned long long> done{0};                                                            \
    if (!((done.load(std::memory_order_relaxed) >> dev) & 1ull)) {                                             \
      VT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));        \
      done.fetch_or(1ull << dev, std::memory_order_relaxed);                                                   \
    }                                                                                                          \
    hipLaunchKernelGGL(kern, grid, block, smem, s, Q, ldq, Qlo, ldqlo, Kt, Klo, klo_tiles_per_seq, Vt, tile_table, seqs, O, ldo, Olo, heads, sl2); \
  } while (0)
  if (HD == 128) {
    if (causal) VT_FAP(true, 128); else VT_FAP(false, 128);
  } else {
    if (causal) VT_FAP(true, 64); else VT_FAP(false, 64);
  }
#undef VT_FAP
  VT_LAUNCH_CHECK();
  return VT_OK;
}

size_t vt_attn_decode_scratch_bytes(int nseq, int heads, int HD, int max_kv_len) {
  const int ntiles = (max_kv_len + 63) / 64;
  int nsplit = (ntiles + 3) / 4;
  nsplit = nsplit < 1 ? 1 : (nsplit > 32 ? 32 : nsplit);
  return (size_t)nseq * heads * nsplit * (HD + 2) * sizeof(float);
}

int vt_attn_decode_launch(const bf16_t* Q, int ldq, const bf16_t* Kt, const bf16_t* Vt, const int* tile_table,
                          const VtAttnSeq* seqs, int nseq, bf16_t* O, int ldo, int heads, int HD, float scale,
                          int max_kv_len, float* scratch, size_t scratch_bytes, hipStream_t s) {
  VT_REQUIRE(Q && Kt && Vt && tile_table && seqs && O && scratch, "vt_attn_decode: null pointer");
  VT_REQUIRE(HD == 64 || HD == 128, "vt_attn_decode: head_dim %d unsupported", HD);
  VT_REQUIRE(max_kv_len > 0, "vt_attn_decode: max_kv_len must be > 0");
  const int ntiles = (max_kv_len + 63) / 64;
  int nsplit = (ntiles + 3) / 4;
  nsplit = nsplit < 1 ? 1 : (nsplit > 32 ? 32 : nsplit);
  const size_t need = (size_t)nseq * heads * nsplit * (HD + 2) * sizeof(float);
  if (scratch_bytes < need) {
    vt_set_error("vt_attn_decode: scratch too small (%zu < %zu)", scratch_bytes, need);
    return VT_ERR_WORKSPACE;
  }
  const float sl2 = scale * 1.4426950408889634f;
  VtProfScope prof(VT_PROF_ATTN_DECODE, 0.0, s);
  dim3 grid(heads, nseq, nsplit), block(256);
  if (HD == 64) {
    hipLaunchKernelGGL((attn_decode_kernel<64>), grid, block, 0, s, Q, ldq, Kt, Vt, tile_table, seqs, heads, sl2, scratch, nsplit);
    hipLaunchKernelGGL((attn_decode_combine_kernel<64>), dim3(heads, nseq), dim3(64), 0, s, scratch, seqs, O, ldo, heads, nsplit);
  } else {
    hipLaunchKernelGGL((attn_decode_kernel<128>), grid, block, 0, s, Q, ldq, Kt, Vt, tile_table, seqs, heads, sl2, scratch, nsplit);
    hipLaunchKernelGGL((attn_decode_combine_kernel<128>), dim3(heads, nseq), dim3(64), 0, s, scratch, seqs, O, ldo, heads, nsplit);
  }
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_attn_decode_fused_launch(const bf16_t* qkv, int ldqkv, int q_col0, int k_col0, int v_col0, bf16_t* Kt, bf16_t* Vt,
                                const int* tile_table, const VtAttnSeq* seqs, int nseq, bf16_t* O, int ldo, int heads, int HD,
                                float scale, const float* rope_cos, const float* rope_sin, const int* positions, hipStream_t s) {
  VT_REQUIRE(qkv && Kt && Vt && tile_table && seqs && O, "vt_attn_decode_fused: null pointer");
  VT_REQUIRE(HD == 64 || HD == 128, "vt_attn_decode_fused: head_dim %d unsupported", HD);
  VT_REQUIRE(nseq > 0 && heads > 0, "vt_attn_decode_fused: empty problem");
  VT_REQUIRE(ldqkv % 8 == 0 && q_col0 % 8 == 0 && k_col0 % 8 == 0 && v_col0 % 8 == 0, "vt_attn_decode_fused: misaligned columns");
  if (rope_cos) VT_REQUIRE(rope_sin && positions, "vt_attn_decode_fused: rope needs sin table and positions");
  const float sl2 = scale * 1.4426950408889634f;
  VtProfScope prof(VT_PROF_ATTN_DECODE, 0.0, s);
  dim3 grid(heads, nseq), block(512);
#define VT_ADF(HDV, RV)                                                                                                          \
  hipLaunchKernelGGL((attn_decode_fused_kernel<HDV, RV>), grid, block, 0, s, qkv, ldqkv, q_col0, k_col0, v_col0, Kt, Vt, tile_table, \
                     seqs, heads, rope_cos, rope_sin, positions, sl2, O, ldo)
  if (HD == 64) {
    if (rope_cos) VT_ADF(64, true); else VT_ADF(64, false);
  } else {
    if (rope_cos) VT_ADF(128, true); else VT_ADF(128, false);
  }
#undef VT_ADF
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_attn_temporal_f32_launch(const float* qkv, int ld, bf16_t* out, bf16_t* out_lo, int B, int T, int N, int heads, hipStream_t s) {
  VT_REQUIRE(qkv && out && out_lo, "vt_attn_temporal_f32: null pointer");
  VT_REQUIRE(T >= 1 && T <= 8, "vt_attn_temporal_f32: T=%d unsupported (1..8)", T);
  VT_REQUIRE(ld >= 3 * heads * 64, "vt_attn_temporal_f32: ld %d too small", ld);
  const long total = (long)B * N * heads;
  dim3 grid((unsigned)((total + 3) / 4)), block(256);
  switch (T) {
#define VT_TC(TV) case TV: hipLaunchKernelGGL((attn_temporal_f32_kernel<TV>), grid, block, 0, s, qkv, ld, out, out_lo, B, N, heads); break;
    VT_TC(1) VT_TC(2) VT_TC(3) VT_TC(4) VT_TC(5) VT_TC(6) VT_TC(7) VT_TC(8)
#undef VT_TC
  }
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_attn_temporal_launch(const bf16_t* qkv, bf16_t* out, int B, int T, int N, int heads, hipStream_t s) {
  VT_REQUIRE(qkv && out, "vt_attn_temporal: null pointer");
  VT_REQUIRE(T >= 1 && T <= 8, "vt_attn_temporal: T=%d unsupported (1..8)", T);
  const long total = (long)B * N * heads;
  dim3 grid((unsigned)((total + 3) / 4)), block(256);
  if (T == 8) {
    hipLaunchKernelGGL(attn_temporal8_kernel, grid, block, 0, s, qkv, out, B, N, heads);
    VT_LAUNCH_CHECK();
    return VT_OK;
  }
  switch (T) {
#define VT_TC(TV) case TV: hipLaunchKernelGGL((attn_temporal_kernel<TV>), grid, block, 0, s, qkv, out, B, N, heads); break;
    VT_TC(1) VT_TC(2) VT_TC(3) VT_TC(4) VT_TC(5) VT_TC(6) VT_TC(7) VT_TC(8)
#undef VT_TC
  }
  VT_LAUNCH_CHECK();
  return VT_OK;
}
