// vt_attn_w4.hip -- flash_attn_w4_kernel: prefill attention (head_dim 128) with ONE wave per SIMD and the K step placed by hand.
//
// Replaces, for long sequences, the two-waves-per-SIMD kernel of vt_attn.hip (flash_attn_kernel<128, *, 4, 2>): there every wave owns
// 32 query rows and reads the WHOLE K and V^T tile from LDS (32 FLOP per LDS byte: the LDS pipe and the matrix pipe are co-limited,
// PMC: MFMA busy 43 %), and the softmax VALU work of one wave only overlaps the other wave's MFMAs by arbitration (13 % co-execution).
// Here (round 4; the structure /opt/skills/guides/cdna_hip_programming.md prices at 1.25-1.40 PFLOP/s, built with the discipline of
// gemm_w4_kernel instead of compiler scheduling):
//   * a workgroup = 4 waves = 256 query rows, a wave = 64 rows = two 32-row halves that SHARE every K / V^T fragment read (half the LDS
//     bytes per FLOP), one wave per SIMD with the whole register file: O^T (128 registers) lives in the accumulator file, pinned by
//     inline-asm MFMAs ("+a"), Q (64) sits there too as the B operand of the score MFMAs ("a"), the arch VGPRs hold two S buffers, two P
//     buffers and three-deep K / V^T fragment windows;
//   * the softmax unit is a 32-key SUB tile and the loop is software-pipelined three deep: in one sub-iteration the matrix pipe runs
//     PV(u) and QK(u+2) (16 + 16 MFMAs of 32 cycles, alternating, so dependent MFMAs are 4 / 16 issues apart) while the vector pipe
//     runs the whole softmax of sub tile u+1 (row max, deferred rescale test, scale-and-shift, exp2, row sum, fp16 pack: ~140 VALU
//     instructions) -- co-issue works for instructions of ONE wave, not between the two waves of a SIMD;
//   * every MFMA gap carries its share of that stream plus the LDS fragment reads (8 + 8 ds_read_b128), the LDS-DMA of the tile three
//     ahead (8 pieces through a buffer resource built once per tile) and the address updates: placed gap by gap by
//     tools/gen_attn_w4.py (-> vt_attn_w4_si0.inc / vt_attn_w4_si1.inc), fenced with sched_barrier; the VALU work goes through
//     single-instruction asm helpers (no SLP packing into v_pk_*, no canonicalising v_max in front of the MFMA outputs);
//   * K / V^T tiles arrive by LDS-DMA into a four-deep ring (K ring [0, 64 KiB), V^T ring [64, 128 KiB)), K of tile T+3 first and
//     waited for at the end of the tile (counted vmcnt), V^T of T+3 one tile later; ONE barrier per 64-key tile;
//   * deferred rescale as in the other kernel (threshold 2^8, fp16 P with exponent bias 2^7); the rare O rescale is applied between
//     two sub-iterations so that it never splits a pending P.V product.
// Same tile / page layout, same permuted K rows, same results contract as flash_attn_kernel (tests compare the two kernels).
// PLACED = false keeps the same pipeline with the three stages run one after the other by simple loops: the reference the placed
// schedule is tested against, and the prologue / tail of the placed kernel.
// Around the kernel (host side, end of this file): causal launches of more than one round hand their blocks to the dispatcher in a
// PLANNED order (vt_flash_attn_w4_block_order: the order is the schedule); PERSIST = true is the persistent form (one workgroup per CU
// walking the block list, K / V^T stream running on across block seams) -- correct, selectable, slower today.
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "vt_common.h"
#include "vt_kernels.h"
#include "vt_mx4.h"

namespace {

#define W4A_THR 8.0f
#define W4A_BIAS 7.0f

// single-instruction VALU helpers (see header)
__device__ __forceinline__ float a_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float a_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (fma / exp2 / add / pack are compiler-visible instructions, not asm: the hazard recognizer counts an inline-asm statement as ZERO wait
// states, so a chain of asm helpers gets an s_nop between every dependent pair -- one per exponential. This file is built with
// -fno-slp-vectorize (vitron_amd/build.py) so that adjacent scalar fmas / adds are not packed into v_pk_*_f32, an anti-lever beside
// MFMAs. v_max3 stays asm: fmaxf on values the compiler cannot prove canonical costs an extra v_max per input.)
__device__ __forceinline__ float a_fma_s(float s, float scale_sgpr, float add) { return __builtin_fmaf(s, scale_sgpr, add); }
__device__ __forceinline__ float a_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float a_add(float a, float b) { return a + b; }
__device__ __forceinline__ uint32_t a_cvt_pk_f16(float lo, float hi) { return pack_f16x2(lo, hi); }
// max of a value with the other half-wave's lane (lane ^ 32): v_permlane32_swap exchanges the upper half of its first operand with
// the lower half of its second; with both holding x the results are (x_lo, x_lo) and (x_hi, x_hi), and their max is the row's.
// Spelled in asm: this toolchain's __builtin_amdgcn_permlane32_swap hands back its FIRST result for both elements of the pair it
// returns (checked in the ISA: the max of the halves silently became max(own half, own half), harmless on mild scores because of the
// deferred-rescale slack, overflowing the fp16 softmax weights on wide ones). s_nop 1: VALU write -> permlane read.
__device__ __forceinline__ float a_max_halves(float x) {
  float t, y = x;
  asm("v_mov_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "=&v"(t), "+v"(y));
  return t;
}

#define W4A_QK0(D, A, B) asm volatile(VT_MFMA_32x32x16_ASM " %0, %1, %2, 0" : "=&v"(D) : "v"(A), "a"(B))
#define W4A_QK(D, A, B) asm volatile(VT_MFMA_32x32x16_ASM " %0, %1, %2, %0" : "+v"(D) : "v"(A), "a"(B))
#define W4A_PV(D, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(D) : "v"(A), "v"(B))
#define W4A_LDS(p) ((__attribute__((address_space(3))) void*)(p))

// Ticket counters of the persistent form, one per XCD; self-resetting: every launch performs exactly (blocks + workgroups) increments
// per counter with atomicInc's wrap value set to that count - 1, so the counter is back at 0 when the launch retires. One launch of
// this kernel at a time per operand library (launches of one stream are ordered; do not run it on two streams concurrently).
__device__ unsigned g_w4_ticket[8];
#ifdef VT_ABLATIONS   // phase clocks of the persistent form (test library only: VT_W4_PROF=1 python tools/attn_bench.py prints them)
__device__ unsigned long long g_w4_prof[4][16];
#define W4A_PROF(I)                                                                                                     \
  do {                                                                                                                  \
    if ((PERSIST ? blockIdx.x == 8 : (blockIdx.x == 8 && blockIdx.y == 0 && blockIdx.z == 0)) && lane == 0) {          \
      const unsigned long long now_ = __builtin_readcyclecounter();                                                    \
      g_w4_prof[wave][I] += now_ - prof_t;                                                                              \
      prof_t = now_;                                                                                                    \
    }                                                                                                                   \
  } while (0)
#else
#define W4A_PROF(I) do {} while (0)
#endif

// ABL (timing ablations of the placed loop, results are garbage; test library only): 1 = no LDS-DMA, 2 = no exp phase of the softmax,
// 4 = no fragment reads, 8 = no end-of-tile waits / barrier, 16 = no P.V MFMAs, 32 = no score MFMAs
// PERSIST: gridDim.x workgroups (one per CU) walk the (sequence, block, head) list instead of one workgroup per block:
//   * workgroup w serves XCD x = w % nx (nx = 8 when heads % 8 == 0: a head's K / V^T pages are then only ever read through ONE L2) and
//     takes tickets from g_w4_ticket[x]; ticket j -> block y = j / (hpx * nseq) counted from the heaviest (last) causal block, sequence
//     z = (j / hpx) % nseq, head = (j % hpx) * nx + x. Wave 0 takes the ticket two blocks ahead (the atomic is issued before a wait the
//     block start performs anyway) and hands it to the other waves through one LDS word;
//   * the K / V^T stream runs on ACROSS the block seam: a block whose tile count is a multiple of 4 (every causal prefill block of a
//     sequence without past) ends with its last tile in ring slot 3, so the steady-state bodies that consume tiles n-3 and n-2 fetch
//     tiles 0 and 1 of the NEXT block into slots 0 and 1 (where they belong), the tail fetches its tile 2 and loads its Q rows into the
//     registers the last score MFMA has just released, and O is staged in the two slots the tail frees (K slot 3, V^T slot 3) -- the next
//     block starts with its operands in flight or landed instead of cold. Other tile counts fall back to a cold start per block.
// MXOUT (precise level 3): the epilogue also writes the MX-FP4 image of the output's rounding remainder (O4 / oexp: o_proj's second operand,
// vt_mx4.h); the other instantiations never read the two pointers.
template <bool CAUSAL, bool PLACED, int ABL = 0, bool PERSIST = false, bool MXOUT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void flash_attn_w4_kernel(
    const op16_t* __restrict__ Q, int ldq, const op16_t* __restrict__ Kt, const op16_t* __restrict__ Vt,
    const int* __restrict__ tile_table, const VtAttnSeq* __restrict__ seqs, op16_t* __restrict__ O, int ldo, int heads,
    float scale_log2e, int nqb_max, int nseq, int nx, const int* __restrict__ order, uint8_t* __restrict__ O4, uint8_t* __restrict__ oexp) {
  constexpr int HD = 128, QBLK = 256, TB = 64 * HD * 2, VRING = 4 * TB;   // 16-KiB tiles; K ring then V^T ring, 4 slots each
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // every pointer argument is asked for NOW (the launcher has checked them): left to itself the compiler fetches the descriptor pointer
  // first and the rest of the argument block only behind the descriptor's own load -- one more scalar round trip in front of the first
  // DMA of every block. (Not an asm statement: a volatile asm in front of the descriptor load turns it into a VECTOR load -- the
  // "not clobbered" proof scalar loads need fails -- and everything derived from it into divergent values.)
  if (__builtin_expect(!Q || !Kt || !Vt || !tile_table || !seqs || !O || heads <= 0 || ldq <= 0 || ldo <= 0, 0)) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ql = lane & 31, hh = lane >> 5;

  // ---- the block this workgroup works on (wave-uniform; re-assigned per block in the persistent form) ---------------------------------
  VtAttnSeq sq;
  int head = 0, qb = 0, past = 0, ntiles = 0, wrow0 = 0;
  const size_t tile_stride = (size_t)heads * 64 * HD;
  auto tiles_of = [&](const VtAttnSeq& s, int qb_) {
    int n = (s.kv_len + 63) >> 6;
    if (CAUSAL) {
      const int last_key = (s.kv_len - s.q_len) + min(qb_ * QBLK + QBLK - 1, s.q_len - 1);
      n = min(n, (last_key >> 6) + 1);
    }
    return n;
  };
  auto set_block = [&](const VtAttnSeq& s, int head_, int qb_) {
    sq = s;
    head = head_;
    qb = qb_;
    past = s.kv_len - s.q_len;
    ntiles = tiles_of(s, qb_);
    wrow0 = qb_ * QBLK + wave * 64;          // first row of this wave inside the sequence
  };
  // the block after it (persistent form): what the seam needs of it
  VtAttnSeq nsq = {0, 0, 0, 0};
  int nhead = 0, nqb = 0, n_ntiles = 1;
  bool warm_next = false;   // this block hands its successor a running K / V^T stream
  bool warm_now = false;    // ... and this block was handed one

  // ---- Q fragments (B operand of the score MFMAs): lane (q, h) holds d = ks*16 + h*8 .. +7 of its row, both row halves --------
  bf16x8 qf[2][8];
  auto load_q = [&](const VtAttnSeq& s, int head_, int qb_) {
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      const int row = qb_ * QBLK + wave * 64 + rh * 32 + ql;
      const op16_t* qp = Q + (size_t)(s.q_row0 + min(row, s.q_len - 1)) * ldq + head_ * HD + hh * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) qf[rh][ks] = *(const bf16x8*)(qp + ks * 16);
    }
  };

  // ---- LDS-DMA: piece p = wave*4 + i of a tile covers LDS bytes [p*1024, +1024); swizzle applied on the SOURCE side -------------
  int k_vo[4], v_vo[4];   // byte offsets inside the 16-KiB source tile
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int byte = (wave * 4 + i) * 1024 + lane * 16;
    {
      const int row = byte >> 8, c = (byte & 255) >> 4;            // K rows are 256 B
      k_vo[i] = (row * HD + ((c ^ (row & 15)) << 3)) * 2;
    }
    {
      const int row = byte >> 7, c = (byte & 127) >> 4;            // V^T rows are 64 keys = 128 B
      v_vo[i] = (row * 64 + ((c ^ ((row >> 1) & 7)) << 3)) * 2;
    }
  }
  // (the pool can exceed 4 GiB: the buffer resource is rebuilt per tile around the tile's own 16 KiB -- scalar work only)
#define W4A_RSRC_OFF(BASE, OFF) __builtin_amdgcn_make_buffer_rsrc((void*)((BASE) + (OFF)), 0, TB, 0x00020000)
#define W4A_DMA_K(R, SLOT, I) __builtin_amdgcn_raw_ptr_buffer_load_lds(R, W4A_LDS(smem + (SLOT) * TB + (wave * 4 + (I)) * 1024), 16, k_vo[I], 0, 0, 0)
#define W4A_DMA_V(R, SLOT, I) __builtin_amdgcn_raw_ptr_buffer_load_lds(R, W4A_LDS(smem + VRING + (SLOT) * TB + (wave * 4 + (I)) * 1024), 16, v_vo[I], 0, 0, 0)
  // page and head offset of tile i of the K / V^T stream: tile i of this block, or -- past its end, when the stream runs on -- tile
  // i - ntiles of the next one; otherwise the last tile again (a harmless re-fetch into a slot nobody reads any more). The page is a
  // scalar load: callers take it one body ahead of its first use.
  auto stream_page = [&](int i, int& pg_, size_t& ho_) {
    const bool nx_ = warm_next && i >= ntiles;
    const int a = sq.table_off + min(i, ntiles - 1), b = nsq.table_off + min(i - ntiles, n_ntiles - 1);
    const int ha = head, hb = nhead;
    pg_ = tile_table[nx_ ? b : a];
    ho_ = (size_t)(nx_ ? hb : ha) * (64 * HD);
  };
  auto stream_off = [&](int i) -> size_t {
    int pg_;
    size_t ho_;
    stream_page(i, pg_, ho_);
    return (size_t)pg_ * tile_stride + ho_;
  };

  // ---- fragment read offsets (bytes inside a ring slot) ------------------------------------------------------------------------------
  // K: MFMA row i = lane&31 reads key pi(i) of the 32-key sub tile (so that a lane's 16 scores are 16 consecutive keys), chunk ks*2 + h
  const int pi = 16 * ((ql >> 2) & 1) + (ql & 3) + 4 * (ql >> 3);
  int kfix[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kfix[ks] = pi * 256 + (((ks * 2 + hh) ^ (pi & 15)) << 4);
  // V^T: MFMA row i reads d = db*32 + i, chunk sub*4 + 2h + j
  int vfix[2][2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int j = 0; j < 2; ++j) vfix[sub][j] = VRING + ql * 128 + (((sub * 4 + 2 * hh + j) ^ ((ql >> 1) & 7)) << 4);

  // ---- state ---------------------------------------------------------------------------------------------------------------------------
  f32x16 oacc[2][4];   // [row half][32-wide d block], accumulator file
  f32x16 sacc[2][2];   // [S buffer = sub][row half]
  u32x4 pfr[2][2][2];  // [P buffer = sub][row half][j]: fp16 softmax weights, B operand of the P.V MFMAs
  float mrun[2], lrun[2];
  float mb[2];                                   // P_BIAS - (m_run or 0)
  float mthr[2];                                 // (m_run + threshold) / scale_log2e: what a RAW score is compared with
  float apend[2];                                // O rescale decided in the last softmax, applied before the next P.V
  bool pend = false;

  // a wait that the accumulator reads / writes around it cannot be moved across
#define W4A_SETTLE_O()                                                                                                              \
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15"                                                                                   \
               : "+a"(oacc[0][0]), "+a"(oacc[0][1]), "+a"(oacc[0][2]), "+a"(oacc[0][3]), "+a"(oacc[1][0]), "+a"(oacc[1][1]),       \
                 "+a"(oacc[1][2]), "+a"(oacc[1][3]))
  // ---- the three stages as plain loops (prologue, tail, and the whole loop when !PLACED) ---------------------------------------------
  auto qk_stage = [&](f32x16(&S)[2], int slot, int sub) {
    const char* kb = smem + slot * TB + sub * 8192;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 kf = *(const bf16x8*)(kb + kfix[ks]);
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        if (ks == 0) W4A_QK0(S[rh], kf, qf[rh][0]);
        else W4A_QK(S[rh], kf, qf[rh][ks]);
      }
    }
    // MFMA results -> VALU: the compiler cannot see that the asm above is an MFMA, and the softmax of this stage-by-stage form is
    // compiler-visible arithmetic that it is free to hoist right behind it; the wait takes S as operands so that nothing crosses it.
    // (The placed loop consumes S a whole sub-iteration later, through asm blocks in program order.)
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(S[0]), "+v"(S[1]));
  };
  auto pv_stage = [&](u32x4(&P)[2][2], int slot, int sub) {
    const char* vb = smem + slot * TB;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const f16x8 vf = *(const f16x8*)(vb + db * 4096 + vfix[sub][j]);
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) W4A_PV(oacc[rh][db], vf, P[rh][j]);
      }
  };
  // O *= alpha of the last rescale decision: between two sub-iterations, i.e. after every issued P.V product and before the next
  auto apply_pending = [&]() {
    if (__builtin_expect(pend, 0)) {      // (cold: laid out behind the loop, the hot path falls through)
      // last MFMA results -> v_accvgpr_read: the compiler cannot see the hazard, and a bare s_nop statement does not hold the reads
      // back either (they have no dependence on it and were hoisted above it): the wait takes the accumulators as operands
      W4A_SETTLE_O();
#pragma unroll
      for (int rh = 0; rh < 2; ++rh)
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[rh][db][r] *= apend[rh];
      W4A_SETTLE_O();                                                   // v_accvgpr_write -> MFMA reading it as C
      pend = false;
    }
  };
  // mask of sub tile (T, sub) -- only tiles that touch the diagonal or the end of the keys (wave-uniform test)
  auto mask_stage = [&](f32x16(&S)[2], int key0) {
    const bool need = (key0 + 32 > sq.kv_len) || (CAUSAL && (key0 + 31 > past + wrow0));
    if (__builtin_expect(need, 0)) {
      // a REAL branch (if-converted this costs 3 VALU per score on every tile). In the placed loop the scores of (T+1, sub 0) come out of
      // MFMAs issued a few instructions before this point and the compiler cannot see that (asm): without the wait the selects below read
      // the buffer's PREVIOUS contents for the keys they keep (seen once the persistent form changed the instruction distance: rows lost
      // accuracy in proportion to their visible keys of the diagonal sub tile). Taken on diagonal / last tiles only.
      asm volatile("s_nop 15\n\ts_nop 15" : "+v"(S[0]), "+v"(S[1]));
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        const int lim = CAUSAL ? min(sq.kv_len - 1, past + wrow0 + rh * 32 + ql) : (sq.kv_len - 1);   // last visible key
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + 16 * hh + r > lim) S[rh][r] = -INFINITY;
      }
    }
  };
  // rare path of the softmax: the running maximum moves (decided wave-wide), l follows at once, O at the next apply_pending()
  // (the common path only asks "does any half row's raw maximum exceed its threshold": two compares and a scalar or; the half-wave
  // exchange that makes the row maximum, and the scaling, happen here)
  const float inv_scale = 1.0f / scale_log2e;
  auto rescale = [&](const float(&mxh)[2]) {
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      const float m_new = fmaxf(mrun[rh], a_max_halves(mxh[rh]) * scale_log2e);
      const float alpha = (m_new == -INFINITY) ? 1.f : fast_exp2(mrun[rh] - m_new);   // m_run = -inf -> 0 (l and O are 0 then)
      mrun[rh] = m_new;
      lrun[rh] *= alpha;
      apend[rh] = pend ? apend[rh] * alpha : alpha;
      mb[rh] = W4A_BIAS - ((m_new == -INFINITY) ? 0.f : m_new);
      mthr[rh] = (m_new + W4A_THR) * inv_scale;
    }
    pend = true;
  };
  auto sm_stage = [&](f32x16(&S)[2], u32x4(&P)[2][2], int key0) {
    mask_stage(S, key0);
    float mxh[2];
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      float mx = a_max3(S[rh][0], S[rh][1], S[rh][2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = a_max3(mx, S[rh][r], S[rh][r + 1]);
      mxh[rh] = a_max(mx, S[rh][15]);
    }
    if (__builtin_expect((__builtin_amdgcn_ballot_w64(mxh[0] > mthr[0]) | __builtin_amdgcn_ballot_w64(mxh[1] > mthr[1])) != 0, 0)) rescale(mxh);
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float e0 = a_exp2(a_fma_s(S[rh][8 * j + 2 * w], scale_log2e, mb[rh]));
          const float e1 = a_exp2(a_fma_s(S[rh][8 * j + 2 * w + 1], scale_log2e, mb[rh]));
          ps0 = a_add(ps0, e0);
          ps1 = a_add(ps1, e1);
          P[rh][j][w] = a_cvt_pk_f16(e0, e1);
        }
      lrun[rh] += ps0 + ps1;
    }
  };

  // ---- which block(s) ----------------------------------------------------------------------------------------------------------------
  // persistent form: tickets (see the header). t_cur / t_nxt are known to every wave; the one after them travels through tick[0].
  volatile unsigned* tick = (volatile unsigned*)(smem + 8 * TB);
  const int xcd = PERSIST ? (int)(blockIdx.x % (unsigned)nx) : 0;
  const int hpx = PERSIST ? heads / nx : heads;
  const unsigned nblk = PERSIST ? (unsigned)(hpx * nseq * nqb_max) : 0u;
  const unsigned nwg_x = PERSIST ? (gridDim.x + (unsigned)(nx - 1 - xcd)) / (unsigned)nx : 0u;
  unsigned t_cur = 0, t_nxt = 0;
  unsigned t_grab = 0;      // (per lane: lane 0 of wave 0 holds the returned value)
  bool grab_out = false;    // wave 0: an atomic is in flight in t_grab
  auto ticket_block = [&](unsigned t, int& z_, int& head_, int& y_) {
    const int per_y = hpx * nseq;
    y_ = (int)t / per_y;
    const int r = (int)t % per_y;
    z_ = r / hpx;
    head_ = (r % hpx) * nx + xcd;
  };
  auto ticket_valid = [&](unsigned t) {        // a block of the grid that exists for its sequence (shorter sequences have fewer blocks)
    if (t >= nblk) return true;                // ... or the end marker
    int z_, h_, y_;
    ticket_block(t, z_, h_, y_);
    return y_ < (seqs[z_].q_len + QBLK - 1) / QBLK;
  };
  auto grab_issue = [&]() {                    // (wave 0 only) ONE increment per wave; the value comes back in lane 0's t_grab
    if (lane == 0) t_grab = atomicInc(&g_w4_ticket[xcd], nblk + nwg_x - 1u);
  };
  auto grab_take = [&]() -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)t_grab); };
  auto grab_valid = [&](bool issued) -> unsigned {   // (wave 0 only) the next ticket that names a real block, or the end marker
    if (!issued) grab_issue();
    unsigned t = grab_take();
    while (!ticket_valid(t)) {
      grab_issue();
      t = grab_take();
    }
    return t;
  };
  if constexpr (PERSIST) {
    if (wave == 0) {
      const unsigned a = grab_valid(false);
      const unsigned b = a < nblk ? grab_valid(false) : a;
      if (lane == 0) {
        tick[1] = a;
        tick[2] = b;
      }
      if (b < nblk) {
        grab_issue();
        grab_out = true;
      }
    }
    __syncthreads();
    t_cur = (unsigned)__builtin_amdgcn_readfirstlane((int)tick[1]);
    t_nxt = (unsigned)__builtin_amdgcn_readfirstlane((int)tick[2]);
  } else {
    // which block: the grid itself (head fastest, heaviest causal blocks of all heads first), or -- causal launches of more than one
    // round -- the dispatch order the launcher planned (w4_block_order below): entry i names the block of workgroup i
    int bx = (int)blockIdx.x, by = (int)blockIdx.y, bz = (int)blockIdx.z;
    if (order) {
      const int lin = order[blockIdx.x];
      bx = lin % heads;
      by = (lin / heads) % nqb_max;
      bz = lin / (heads * nqb_max);
    }
    const VtAttnSeq s0 = seqs[bz];
    const int qb0 = (s0.q_len + QBLK - 1) / QBLK - 1 - by;
    if (qb0 < 0) return;
    set_block(s0, bx, qb0);
  }

  unsigned long long prof_t = __builtin_readcyclecounter();
  (void)prof_t;
  for (;;) {
    W4A_PROF(0);      // seam + whatever preceded the loop top
    if constexpr (PERSIST) {
      if (t_cur >= nblk) break;
      if (!warm_now) {                                   // (a warm block was set up at the seam that handed it over)
        int z_, h_, y_;
        ticket_block(t_cur, z_, h_, y_);
        const VtAttnSeq s_ = seqs[z_];
        set_block(s_, h_, (s_.q_len + QBLK - 1) / QBLK - 1 - y_);
      }
      warm_next = false;
      if (t_nxt < nblk) {
        int z_, y_;
        ticket_block(t_nxt, z_, nhead, y_);
        nsq = seqs[z_];
        nqb = (nsq.q_len + QBLK - 1) / QBLK - 1 - y_;
        n_ntiles = tiles_of(nsq, nqb);
        warm_next = PLACED && (ntiles & 3) == 0 && ntiles >= 4;
      }
    }
    // ---- cold start: Q rows and tiles 0..2 requested now (a warm block got them from its predecessor) --------------------------------
    if (!warm_now) {
      load_q(sq, head, qb);
      // (the three page look-ups go out together: taken one by one, each is a scalar round trip in front of its tile's DMA -- the phase
      // clocks of the first version showed 7.7-10 k cycles of block set-up, three quarters of it these waits)
      int pg3[3];
      {
        const int* tp = tile_table + sq.table_off;
        if (ntiles >= 3) {                               // consecutive entries, no clamping: three loads, one wait
          pg3[0] = tp[0];
          pg3[1] = tp[1];
          pg3[2] = tp[2];
        } else {
          pg3[0] = tp[0];
          pg3[1] = pg3[2] = tp[ntiles - 1];              // (the last tile again)
        }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const size_t off = (size_t)pg3[t] * tile_stride + (size_t)head * (64 * HD);
        const __amdgpu_buffer_rsrc_t rk = W4A_RSRC_OFF(Kt, off), rv = W4A_RSRC_OFF(Vt, off);
#pragma unroll
        for (int i = 0; i < 4; ++i) W4A_DMA_K(rk, t, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) W4A_DMA_V(rv, t, i);
      }
    }
    W4A_PROF(1);      // block set-up, cold start issue
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W4A_PROF(2);      // wait for the operands
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    W4A_PROF(3);      // barrier
    if constexpr (PERSIST) {
      // Q arrives from two program points (cold start above, the previous block's tail): pin it to the accumulator file HERE, once per
      // block -- otherwise the register allocator carries some fragments in VGPRs and copies them (v_accvgpr_write) straight in front
      // of the asm MFMAs that read them, a hazard it cannot see (the MFMA then reads the register's previous contents)
#pragma unroll
      for (int rh = 0; rh < 2; ++rh)
        asm volatile("" : "+a"(qf[rh][0]), "+a"(qf[rh][1]), "+a"(qf[rh][2]), "+a"(qf[rh][3]), "+a"(qf[rh][4]), "+a"(qf[rh][5]), "+a"(qf[rh][6]), "+a"(qf[rh][7]));
      // the ticket two blocks ahead: its atomic went out before the wait above; every wave has read the previous one by now
      if (wave == 0) {
        unsigned t = nblk;
        if (grab_out) {
          t = grab_valid(true);
          grab_out = false;
        }
        if (lane == 0) tick[0] = t;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[rh][db][r] = 0.f;
      mrun[rh] = -INFINITY;
      lrun[rh] = 0.f;
      mb[rh] = W4A_BIAS;
      mthr[rh] = -INFINITY;
      apend[rh] = 1.f;
    }
    pend = false;

    W4A_PROF(4);      // ticket hand-over (wave 0), state reset
    // ---- prologue: scores of tile 0, softmax of its first half -----------------------------------------------------------------------
    qk_stage(sacc[0], 0, 0);
    qk_stage(sacc[1], 0, 1);
    sm_stage(sacc[0], pfr[0], 0);

    // fragment windows of the placed loop: K of k-step ks in kfr[ks % 3], V^T fragment p = (j, db) in vfr[p % 3]; the first three of
    // each are read one sub-iteration ahead (here: K of tile 1 / sub 0 -- a re-fetched copy when there is no tile 1, never used then --
    // and V^T of tile 0 / sub 0)
    bf16x8 kfr[3];
    f16x8 vfr[3];
    if constexpr (PLACED) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        kfr[i] = *(const bf16x8*)(smem + TB + kfix[i]);
        vfr[i] = *(const f16x8*)(smem + i * 4096 + vfix[0][0]);
      }
    }
    W4A_PROF(5);      // prologue stages
    // ---- steady state: one 64-key tile per iteration = two sub-iterations, each  P.V(u) || softmax(u+1) || Q.K^T(u+2) ---------------
    int T = 0;
    int pg;                               // page of the tile the next body fetches: looked up one body ahead (a scalar load at the top of
    size_t pg_head_off;                   // the body would make the wave wait for it AND for every fragment read in flight)
    stream_page(3, pg, pg_head_off);
    // One body per ring slot: the loop is unrolled four times so that every LDS offset of a body (fragment reads, DMA destinations) is an
    // instruction immediate -- with run-time slots each of the 32 fragment reads of a tile cost a scalar add and a vector add on top
    // (PMC of the first version: 8.1 instructions per MFMA, issue-bound; the MFMA itself takes about three of the eight issue slots of
    // its 32 cycles). The body is vt_attn_w4_body.inc, included once per slot.
    if constexpr (PLACED) {
      for (;;) {
        if (T + 1 >= ntiles) break;
#define W4A_SC 0
#include "vt_attn_w4_body.inc"
#undef W4A_SC
        if (T + 1 >= ntiles) break;
#define W4A_SC 1
#include "vt_attn_w4_body.inc"
#undef W4A_SC
        if (T + 1 >= ntiles) break;
#define W4A_SC 2
#include "vt_attn_w4_body.inc"
#undef W4A_SC
        if (T + 1 >= ntiles) break;
#define W4A_SC 3
#include "vt_attn_w4_body.inc"
#undef W4A_SC
      }
    } else {
      // the reference form: stages one after the other, run-time ring slots, uncounted waits -- kept as simple as possible
      for (; T + 1 < ntiles; ++T) {
        const int slot_c = T & 3, slot_n = (T + 1) & 3, slot_d = (T + 3) & 3;
        const size_t off = stream_off(T + 3);                                             // past the end: the last tile again, harmless
        const __amdgpu_buffer_rsrc_t rk = W4A_RSRC_OFF(Kt, off), rv = W4A_RSRC_OFF(Vt, off);
#pragma unroll
        for (int i = 0; i < 4; ++i) W4A_DMA_K(rk, slot_d, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) W4A_DMA_V(rv, slot_d, i);
        apply_pending();
        pv_stage(pfr[0], slot_c, 0);
        sm_stage(sacc[1], pfr[1], T * 64 + 32);
        qk_stage(sacc[0], slot_n, 0);
        apply_pending();
        pv_stage(pfr[1], slot_c, 1);
        sm_stage(sacc[0], pfr[0], (T + 1) * 64);
        qk_stage(sacc[1], slot_n, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    W4A_PROF(6);      // steady state
    // ---- tail: the last tile's second half and both P.V products -----------------------------------------------------------------------
    if constexpr (PERSIST) {
      if (warm_next) {
        // the seam: Q of the next block into the registers the last score MFMA released, its tile 2 into slot 2 (tile n-2's: every
        // wave is behind the barrier that ended the body which read it); tiles 0 and 1 were fetched by the bodies of tiles n-3 and n-2
        load_q(nsq, nhead, nqb);
        const size_t off = stream_off(ntiles + 2);
        const __amdgpu_buffer_rsrc_t rk = W4A_RSRC_OFF(Kt, off), rv = W4A_RSRC_OFF(Vt, off);
#pragma unroll
        for (int i = 0; i < 4; ++i) W4A_DMA_K(rk, 2, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) W4A_DMA_V(rv, 2, i);
      }
    }
    {
      const int slot_c = T & 3;
      apply_pending();
      pv_stage(pfr[0], slot_c, 0);
      sm_stage(sacc[1], pfr[1], T * 64 + 32);
      apply_pending();
      pv_stage(pfr[1], slot_c, 1);
    }
    if (!warm_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail's redundant pieces (none when the stream runs on)
    W4A_SETTLE_O();                                                    // last MFMA results -> the epilogue's v_accvgpr_read

    // ---- epilogue: O / l, packed to 16 bit, staged through LDS and stored as whole rows -----------------------------------------------
    // A lane holds 4 consecutive d (8 bytes) of ITS row per (d block, group): stored straight from the registers that is 32 eight-byte
    // stores per lane, each instruction touching 32 rows -- store-ISSUE-bound (the guide prices such a tail at ~9k cycles per block; with
    // one workgroup per CU nothing overlaps it). Instead every wave parks 32 rows x 256 B at a time in the two ring slots the tail has
    // released -- K slot 3 (waves 0, 1) and V^T slot 3 (waves 2, 3): with the next block's tiles landing in slots 0..2 these are the only
    // free ones; the barrier below is behind everybody's last read of them AND (vmcnt(0) above, cold case) behind everybody's last
    // LDS-DMA -- 16-byte chunks XOR-swizzled by the row, and stores 16 bytes per lane: one instruction = 4 rows of 256 contiguous bytes.
    W4A_PROF(7);      // tail
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");      // (the barrier builtin is not a compiler fence for LDS accesses)
    W4A_PROF(8);      // barrier
    char* stage = smem + (wave < 2 ? 3 * TB : VRING + 3 * TB) + (wave & 1) * 8192;
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      const float l_tot = lrun[rh] + __shfl_xor(lrun[rh], 32, 64);
      const float inv = (l_tot > 0.f) ? 1.f / l_tot : 0.f;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        float lo[MXOUT ? 16 : 1];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float a0 = oacc[rh][db][4 * g + 0] * inv, a1 = oacc[rh][db][4 * g + 1] * inv, a2 = oacc[rh][db][4 * g + 2] * inv,
                      a3 = oacc[rh][db][4 * g + 3] * inv;
          u32x2 o;
          o.x = pack_op2(a0, a1);
          o.y = pack_op2(a2, a3);
          *(u32x2*)(stage + ql * 256 + (((db * 4 + g) ^ (ql & 15)) << 4) + hh * 8) = o;      // d = db*32 + 8g + 4hh .. +3
          if constexpr (MXOUT) {
            lo[4 * g + 0] = a0 - oplo_to_f32(o.x);
            lo[4 * g + 1] = a1 - ophi_to_f32(o.x);
            lo[4 * g + 2] = a2 - oplo_to_f32(o.y);
            lo[4 * g + 3] = a3 - ophi_to_f32(o.y);
          }
        }
        if constexpr (MXOUT) {   // the remainder's 4-bit image straight from the registers: 8 bytes per lane and block
          const int rq = wrow0 + rh * 32 + ql;
          const int m = sq.q_row0 + min(rq, sq.q_len - 1);
          mx4_store_attn_block(lo, hh, rq < sq.q_len, O4 + (size_t)m * (heads * 64) + head * 64 + db * 16,
                               oexp + mx4_aexp_index(m, head * 4 + db, heads * 4));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own writes (the region is wave-private: no barrier)
      const int c = lane & 15, r4 = lane >> 4;               // 16-byte chunk of the row, row inside a group of 4
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rl = i * 4 + r4, row = rh * 32 + rl;
        const u32x4 v = *(const u32x4*)(stage + rl * 256 + ((c ^ (rl & 15)) << 4));
        if (wrow0 + row < sq.q_len) *(u32x4*)(O + (size_t)(sq.q_row0 + wrow0 + row) * ldo + head * HD + c * 8) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // ... read back before the second half overwrites them
    }
    W4A_PROF(9);      // epilogue
#ifdef VT_ABLATIONS
    if (!PERSIST && blockIdx.x == 8 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
      g_w4_prof[wave][10] += 1;
      g_w4_prof[wave][11] += (unsigned long long)ntiles;
    }
#endif
    if constexpr (!PERSIST) break;
    // ---- seam ---------------------------------------------------------------------------------------------------------------------------
    {
      const unsigned t_nn = (unsigned)__builtin_amdgcn_readfirstlane((int)tick[0]);   // written by wave 0 behind this block's first barrier, read behind its last
      if (wave == 0 && t_nn < nblk) {            // the ticket after that one: out now, picked up behind the next block's first wait
        grab_issue();
        grab_out = true;
      }
#ifdef VT_ABLATIONS
      if (blockIdx.x == 8 && lane == 0) {
        g_w4_prof[wave][10] += 1;
        g_w4_prof[wave][11] += (unsigned long long)ntiles;
        g_w4_prof[wave][12] += warm_next ? 1 : 0;
      }
#endif
      t_cur = t_nxt;
      t_nxt = t_nn;
      warm_now = warm_next;
      if (warm_now) set_block(nsq, nhead, nqb);
    }
  }
}

}  // namespace

#ifdef VT_ABLATIONS
extern "C" int vt_debug_w4_prof(unsigned long long* out, int reset) {   // (test library only, not in the header)
  if (out) VT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w4_prof), sizeof(unsigned long long) * 64));
  if (reset) {
    static unsigned long long z[64];
    VT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_w4_prof), z, sizeof(z)));
  }
  return VT_OK;
}
#endif

// ------------------------------------------------------------------------------------------------------------------
// Dispatch order of a causal launch (host). The hardware hands workgroups to CUs in index order, each to the first CU that frees up
// (list scheduling); with the grid's natural order -- heaviest blocks first -- that is LPT, and for ONE 5120-token sequence (640
// blocks of 4 .. 80 tiles on 256 CUs, 2.5 per CU) LPT ends 11.8 % above the mean load: it spends the 256 largest blocks on 256 CUs
// and stacks the small ones on top of pairs that are already too long. A bin packing at a target makespan (MULTIFIT: first-fit
// decreasing inside a binary search on the capacity) finds the assignment LPT cannot -- the seven largest sizes paired with their
// complements, an eighth of the CUs running nothing but chains of small blocks: 2.9 % above the mean -- and listing the blocks by
// their PLANNED START TIME makes the dispatcher reproduce it (the CU that frees up at a block's planned start is the one it was packed
// on). Planned per XCD (workgroup i runs on XCD i % 8; heads = XCD mod 8 keeps a head's K / V^T pages in one L2), identical plans
// interleaved. Block cost = 21.8 k cycles + 3183 per 64-key tile (phase clocks of the kernel); sequences are taken as
// equally long (their lengths live on the device): the order is a permutation of the grid whatever it is, only the balance depends on
// the model. Returns the plan only where it beats the natural order under the model by more than 2 %.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct W4Job {
  double cost;
  int lin;      // head-group-local linear id: hl + hpx * (y + nqb * z)
};
double w4_list_makespan(const std::vector<W4Job>& order, int m) {
  std::vector<double> cu(m, 0.0);   // (m <= 256: a linear scan per job is cheap enough)
  for (const W4Job& j : order) {
    int best = 0;
    for (int i = 1; i < m; ++i)
      if (cu[i] < cu[best]) best = i;
    cu[best] += j.cost;
  }
  return *std::max_element(cu.begin(), cu.end());
}
bool w4_ffd(const std::vector<W4Job>& sorted, double T, int m, std::vector<std::vector<W4Job>>& bins) {
  bins.clear();
  std::vector<double> load;
  for (const W4Job& j : sorted) {
    size_t b = 0;
    for (; b < bins.size(); ++b)
      if (load[b] + j.cost <= T) break;
    if (b == bins.size()) {
      if ((int)bins.size() >= m) return false;
      bins.emplace_back();
      load.push_back(0.0);
    }
    bins[b].push_back(j);
    load[b] += j.cost;
  }
  return true;
}
}  // namespace

// order[i] = block (head + heads * (y + nqb * z)) of workgroup i; empty = keep the grid's natural order
std::vector<int> vt_flash_attn_w4_block_order(int heads, int nqb, int nseq, int ncu) {
  const int nx = (heads % 8 == 0 && ncu % 8 == 0) ? 8 : 1;
  const int hpx = heads / nx, m = ncu / nx;
  const long n = (long)hpx * nqb * nseq;
  if (n <= m || n > 4096) return {};                 // one round, or so many blocks that list scheduling balances by itself
  double fixed = 21.8e3;                           // per-block cycles outside the tile loop
#ifdef VT_ABLATIONS
  if (getenv("VT_W4_ORDER_FIXED")) fixed = atof(getenv("VT_W4_ORDER_FIXED"));   // (test library: A/B of the cost model)
#endif
  std::vector<W4Job> jobs;
  jobs.reserve(n);
  for (int z = 0; z < nseq; ++z)
    for (int y = 0; y < nqb; ++y)
      for (int hl = 0; hl < hpx; ++hl) jobs.push_back({fixed + 3183.0 * 4 * (nqb - y), hl + hpx * (y + nqb * z)});
  // the natural order, y-major per sequence (what the 3-D grid dispatches) -- and its makespan under the model
  const double natural = w4_list_makespan(jobs, m);
  std::vector<W4Job> sorted = jobs;
  std::stable_sort(sorted.begin(), sorted.end(), [](const W4Job& a, const W4Job& b) { return a.cost > b.cost; });
  double total = 0.0;
  for (const W4Job& j : jobs) total += j.cost;
  double lo = std::max(total / m, sorted[0].cost), hi = 2.0 * lo;
  std::vector<std::vector<W4Job>> bins, best;
  for (int it = 0; it < 24; ++it) {
    const double T = 0.5 * (lo + hi);
    if (w4_ffd(sorted, T, m, bins)) {
      best = bins;
      hi = T;
    } else {
      lo = T;
    }
  }
  if (best.empty()) return {};
  struct Ev {
    double start, cost;
    int lin;
  };
  std::vector<Ev> ev;
  for (const auto& b : best) {
    double t = 0.0;
    for (const W4Job& j : b) {            // (largest first inside a bin: first-fit decreasing filled it that way)
      ev.push_back({t, j.cost, j.lin});
      t += j.cost;
    }
  }
  std::stable_sort(ev.begin(), ev.end(), [](const Ev& a, const Ev& b) { return a.start != b.start ? a.start < b.start : a.cost > b.cost; });
  std::vector<W4Job> planned;
  for (const Ev& e : ev) planned.push_back({e.cost, e.lin});
  if (w4_list_makespan(planned, m) > 0.98 * natural) return {};
  std::vector<int> order((size_t)n * nx);
  for (long p = 0; p < n; ++p)
    for (int x = 0; x < nx; ++x) {
      const int lin = planned[p].lin, hl = lin % hpx, yz = lin / hpx;   // XCD x runs the same plan on its own heads
      order[(size_t)p * nx + x] = (hl * nx + x) + heads * yz;
    }
  return order;
}

bool vt_flash_attn_w4_supported(int HD, int max_q_len, int heads, int nseq) {
  (void)heads;
  (void)nseq;
  return HD == 128 && max_q_len >= 1;
}

int vt_flash_attn_w4_launch(const bf16_t* Q, int ldq, const bf16_t* Kt, const bf16_t* Vt, const int* tile_table, const VtAttnSeq* seqs,
                            int nseq, int max_q_len, bf16_t* O, int ldo, int heads, int causal, float scale_log2e, int placed,
                            hipStream_t s, uint8_t* O4, uint8_t* oexp) {
  VT_REQUIRE(!O4 || (oexp && causal && placed == 1 && ldo == heads * 128), "vt_flash_attn_w4: the level 3 operand out runs on the placed causal kernel, dense output");
  constexpr int smem = 8 * 16384 + 16;   // K ring + V^T ring, four 16-KiB slots each (+ the persistent form's ticket words)
  const int nqb_max = cdiv(max_q_len, 256);
  // placed == 2: the persistent form -- one workgroup per CU walking the block list (kernel header)
  // per-DEVICE state: CU count, the dynamic-LDS attribute of each instantiation, the planned-order buffers
  int dev = 0;
  VT_HIP(hipGetDevice(&dev));
  VT_REQUIRE(dev >= 0 && dev < 64, "vt_flash_attn_w4_launch: device ordinal %d", dev);
  static std::atomic<int> ncu_of[64];
  int ncu = ncu_of[dev].load(std::memory_order_relaxed);
  if (!ncu) {
    VT_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    ncu_of[dev].store(ncu, std::memory_order_relaxed);
  }
  const int nx = (heads % 8 == 0 && ncu % 8 == 0) ? 8 : 1;
  const long nblocks = (long)heads * nseq * nqb_max;
  const long cap = g_vt_flash_attn_wgs > 0 ? std::max(g_vt_flash_attn_wgs / nx, 1) * nx : ncu;
  const int ngrid = (int)std::min<long>(cap, std::max<long>(nblocks / nx, 1) * nx);   // a multiple of nx: every XCD gets the same number of workgroups
#define VT_FAW4(CV, PV_, ...)                                                                                  \
  do {                                                                                                         \
    auto kern = flash_attn_w4_kernel<CV, PV_ __VA_OPT__(,) __VA_ARGS__>;                                                                 \
    static std::atomic<unsigned long long> done{0};                                                            \
    if (!((done.load(std::memory_order_relaxed) >> dev) & 1ull)) {                                             \
      VT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));        \
      done.fetch_or(1ull << dev, std::memory_order_relaxed);                                                   \
    }                                                                                                          \
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, Q, ldq, Kt, Vt, tile_table, seqs, O, ldo, heads, scale_log2e, nqb_max, nseq, nx, order, O4, oexp); \
  } while (0)
  dim3 grid(heads, nqb_max, nseq);
  // causal launches of more than one round: the planned dispatch order (above), computed once per shape and kept on the device
  const int* order = nullptr;
  if (causal && placed != 2 && g_vt_flash_attn_order) {
    // The one documented exception to "never allocates" (include/vitron_hip.h, Conventions): the FIRST launch of a new
    // (device, heads, blocks, sequences) shape plans the order on the host (~0.1 ms), keeps it in a pinned host buffer and a device
    // buffer of heads * blocks * sequences ints that live as long as the process (at most a few KiB per shape; the serving shapes
    // are a handful), and uploads it with hipMemcpyAsync on `s` -- stream-ordered before the launch below, no device
    // synchronisation. A stream capture must therefore see a shape for the second time (warm it once, as for any graph).
    struct OrderBuf { int* dev; int* host; hipEvent_t ready; hipStream_t up; bool landed; };
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, OrderBuf> cache;
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_tuple(dev, heads, nqb_max, nseq);
    auto it = cache.find(key);
    if (it == cache.end()) {
      const std::vector<int> o = vt_flash_attn_w4_block_order(heads, nqb_max, nseq, ncu);
      OrderBuf b{nullptr, nullptr, nullptr, s, false};
      if (!o.empty()) {
        const size_t bytes = o.size() * sizeof(int);
        VT_HIP(hipHostMalloc((void**)&b.host, bytes, hipHostMallocDefault));
        std::memcpy(b.host, o.data(), bytes);
        VT_HIP(hipMalloc((void**)&b.dev, bytes));
        VT_HIP(hipMemcpyAsync(b.dev, b.host, bytes, hipMemcpyHostToDevice, s));
        VT_HIP(hipEventCreateWithFlags(&b.ready, hipEventDisableTiming));
        VT_HIP(hipEventRecord(b.ready, s));
      }
      it = cache.emplace(key, b).first;
    }
    OrderBuf& ob = it->second;
    if (ob.dev && !ob.landed && ob.up != s) {       // another stream than the one that carries the upload: order behind it
      if (hipEventQuery(ob.ready) == hipSuccess) ob.landed = true;
      else VT_HIP(hipStreamWaitEvent(s, ob.ready, 0));
    }
    order = ob.dev;
    if (order) grid = dim3((unsigned)nblocks);
  }
#ifdef VT_ABLATIONS   // VT_W4_ABL=<bits> python tools/attn_bench.py (test library only)
  static const int abl = getenv("VT_W4_ABL") ? atoi(getenv("VT_W4_ABL")) : 0;
  if (placed == 1 && causal && abl) {
    switch (abl) {
      case 1: VT_FAW4(true, true, 1); break;
      case 2: VT_FAW4(true, true, 2); break;
      case 3: VT_FAW4(true, true, 3); break;
      case 4: VT_FAW4(true, true, 4); break;
      case 5: VT_FAW4(true, true, 5); break;
      case 6: VT_FAW4(true, true, 6); break;
      case 7: VT_FAW4(true, true, 7); break;
      case 8: VT_FAW4(true, true, 8); break;
      case 9: VT_FAW4(true, true, 9); break;
      case 15: VT_FAW4(true, true, 15); break;
      case 31: VT_FAW4(true, true, 31); break;
      case 47: VT_FAW4(true, true, 47); break;
      case 63: VT_FAW4(true, true, 63); break;
      case 127: VT_FAW4(true, true, 127); break;
      case 191: VT_FAW4(true, true, 191); break;
      case 255: VT_FAW4(true, true, 255); break;
      default: VT_FAW4(true, true); break;
    }
    VT_LAUNCH_CHECK();
    return VT_OK;
  }
#endif
  if (placed == 2) {
    grid = dim3(ngrid);
    if (causal) VT_FAW4(true, true, 0, true); else VT_FAW4(false, true, 0, true);
  } else if (O4) {
    VT_FAW4(true, true, 0, false, true);
  } else if (placed) {
    if (causal) VT_FAW4(true, true); else VT_FAW4(false, true);
  } else {
    if (causal) VT_FAW4(true, false); else VT_FAW4(false, false);
  }
#undef VT_FAW4
  VT_LAUNCH_CHECK();
  return VT_OK;
}
