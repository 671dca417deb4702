// vt_gemm8.hip -- the large-shape bf16 GEMMs: 256x256x64 (320x256x64) tiles, one workgroup per CU.
//
// Two kernel families share the tile, the LDS image (half tiles, source-side XOR swizzle) and the epilogue semantics:
//   * gemm_w4_kernel (round 2, the DEFAULT for plain-store epilogues and for 320-row tiles): four waves of 128x128 (160x128),
//     one per SIMD, accumulators pinned to the accumulator file, the K step placed by hand -- described at its definition;
//   * the 8-wave ping-pong kernel described first below: the 8-phase schedule (gemm_p8_kernel, P4 = false), the 4-phase schedule
//     that merges its phases pairwise (P4 = true: half the barriers, 32 MFMAs per section, +8..14 % -- its stage schedule and
//     hazard argument sit next to its loop; used for VALU-heavy epilogues on 256-row tiles, the folded RMSNorm, the fused QKV
//     epilogue and the split-K passes), and a register-pipelined variant (gemm_rp_kernel, opt-in).
// The two-pass split-K of the residual epilogue (vt_gemm_p4_splitk_resid_launch) lives here too; which kernel runs a given shape
// is decided by the planner in vt_gemm.hip.
//
// Same contract and epilogues as vt_gemm.hip's tile kernel (C = epi(A[M,K].W[N,K]^T + bias)); used for the
// decoder's big projections where >92 % of the prefill FLOPs live.
//
// Structure (one workgroup = 512 threads = 8 waves as 2(M) x 4(N); two waves per SIMD):
//   * The 256x64 A tile and the 256x64 B tile of one K step are staged as four 16-KiB "half tiles"
//     A0 (rows 0..127), A1 (rows 128..255), B0, B1; LDS holds two K steps = 8 slots = 128 KiB.
//   * Wave (wr, wc) owns four 64x32 output quadrants: rows {wr*64.., 128 + wr*64..} x cols {wc*32.., 128 + wc*32..}.
//     One K step = 4 phases, one quadrant each: Q00 (reads A0, B0), Q01 (reads B1), Q11 (reads A1), Q10 (reads nothing:
//     the B0 fragments stay in registers). So every half tile is needed at a different phase and is dead early.
//   * Each phase also issues the LDS-DMA (global_load_lds_dwordx4) of ONE half tile, 6 phases ahead of its first use:
//     stage of (K step u, slot j) goes out at global phase 4u + j - 6. Waits are counted, never drained:
//     s_waitcnt vmcnt(6) leaves the three newest half tiles in flight across the barriers.
//   * The two wave rows are staggered by one barrier (wr == 1 runs one s_barrier behind), so on every SIMD one wave is
//     in its MFMA section while its partner issues ds_reads / DMA: the matrix pipe and the LDS pipe overlap.
//   * LDS image is lane-linear per DMA piece, chunks XOR-swizzled by ((row>>1)&7) on the source address and on the
//     ds_read_b128 address (conflict-free for the 16x16x32 fragment read pattern).
//
// Safety of the schedule (g = global phase, stage issued in the "load" section L(g), wait in the "MFMA" section M(g)):
//   RAW  a slot staged at L(g) is first read at phase >= g+5. A read in L(p) needs every wave's DMA landed before the
//        barrier that opens L(p): the reader's own group waited at M(p-1), the other group (half a phase behind) at
//        M(p-2); both waits are vmcnt(6) = "everything but the 3 newest stages", i.e. stages <= p-5 for the laggard. OK.
//   WAR  stage (u, j) overwrites K step u-2's slot j, whose last read was at phase 4(u-2) + {0,0,1,2}[j]; the stage is
//        issued at 4(u-2) + j + 2 >= two phases later, after the readers' lgkmcnt wait and two barriers. OK.
#include <algorithm>

#include "vt_common.h"
#include "vt_kernels.h"
#include "vt_mx4.h"

namespace {

struct GemmP8 {
  const bf16_t* A;
  const bf16_t* W;
  void* C;
  const float* bias;
  int M, N, K;
  int lda, ldw, ldc;
  int ksplit;        // > 1 (fp32-out epilogue only): blocks [s*tiles, (s+1)*tiles) compute K range s of every tile into slab s of C
  size_t slab;       // elements between consecutive slabs
  VtGemmNormFuse nf; // folded RMSNorm (tile flavour, vt_kernels.h): row_scale on the consumer side, out_* on the producer side
  VtQkvFuse qf;      // EPI == VT_EPI_QKV_PAGES only
};

__device__ __forceinline__ float gelu_erf8(float x) { return vt_gelu_erf(x); }
__device__ __forceinline__ float quick_gelu8(float x) { return vt_quick_gelu(x); }
__device__ __forceinline__ float silu8(float x) { return vt_silu(x); }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

#define VT_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

constexpr int HALF_BYTES = 128 * 64 * 2;        // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;       // A0 A1 B0 B1
constexpr int SLOT_A0 = 0, SLOT_B0 = 1, SLOT_B1 = 2, SLOT_A1 = 3;   // staging order within a K step
__device__ __forceinline__ constexpr int slot_offset(int slot) {     // LDS placement of a slot inside a buffer
  return (slot == SLOT_A0 ? 0 : slot == SLOT_A1 ? 1 : slot == SLOT_B0 ? 2 : 3) * HALF_BYTES;
}

// ABL (timing ablations, results are garbage): bit0 = no fragment reads, bit1 = no LDS-DMA in the loop, bit2 = no MFMA
// P4 = the same pipeline with the phases merged pairwise (see the loop): half as many barriers, 32 MFMAs per section.
template <int EPI, int ABL = 0, bool P4 = false>
__global__ __launch_bounds__(512, 2) void gemm_p8_kernel(GemmP8 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- tile order (XCD remap + groups of 8 M tiles), as in vt_gemm.hip ---------------------------------------------
  const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
  const int nwg = tiles_m * tiles_n;
  // two-pass split-K (EPI == F32 only): split s of every tile writes its partial product to slab s; a reduce kernel adds them
  const int ksplit = (EPI == VT_EPI_F32) ? p.ksplit : 1;
  const int split = (ksplit > 1) ? (int)blockIdx.x / nwg : 0;
  const int sid = xcd_remap((int)blockIdx.x - split * nwg, nwg);
  const int ku = p.K >> 7;   // K in units of 128 (two K steps), balanced over the splits
  const int u_begin = (int)((long)ku * split / ksplit), u_end = (int)((long)ku * (split + 1) / ksplit);
  const int k_begin = u_begin * 128;
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int first_m = (sid / per_group) * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (sid % per_group) % gsz;
  const int tn = (sid % per_group) / gsz;
  const int bm0 = tm * 256, bn0 = tn * 256;

  // ---- DMA source pointers: slot s, piece i (this wave stages pieces 2*wave + i of every half tile) ------------------
  const int lrow = lane >> 3, lchk = lane & 7;
  const bf16_t* src[4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + lrow;          // row inside the half tile
    const int coff = (lchk ^ ((row >> 1) & 7)) * 8;
    src[SLOT_A0][i] = p.A + (size_t)min(bm0 + row, p.M - 1) * p.lda + coff + k_begin;
    src[SLOT_A1][i] = p.A + (size_t)min(bm0 + 128 + row, p.M - 1) * p.lda + coff + k_begin;
    src[SLOT_B0][i] = p.W + (size_t)min(bn0 + row, p.N - 1) * p.ldw + coff + k_begin;
    src[SLOT_B1][i] = p.W + (size_t)min(bn0 + 128 + row, p.N - 1) * p.ldw + coff + k_begin;
  }
  const int dma_off = wave * 2048;                      // this wave's two 1-KiB pieces inside a half tile

#define STAGE(BUF, SLOT)                                                                        \
  do {                                                                                          \
    if ((ABL & 2) && in_loop) break;                                                            \
    char* _d = smem + (BUF) * BUF_BYTES + slot_offset(SLOT) + dma_off;                          \
    glds16(src[SLOT][0], _d);                                                                   \
    glds16(src[SLOT][1], _d + 1024);                                                            \
    src[SLOT][0] += 64;                                                                         \
    src[SLOT][1] += 64;                                                                         \
  } while (0)

  // ---- fragment read offsets (same swizzle as the DMA side) ------------------------------------------------------------
  const int f = (lane >> 1) & 7;
  const int fo0 = (lane & 15) * 128 + (((lane >> 4) ^ f) << 4);
  const int fo1 = (lane & 15) * 128 + ((((lane >> 4) | 4) ^ f) << 4);
  const int a_row_off = wr * 64 * 128;                  // this wave's 64 rows inside an A half tile
  const int b_row_off = wc * 32 * 128;                  // this wave's 32 cols (= rows of W) inside a B half tile

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[a][b][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  bf16x8 af[4][2], b0f[2][2], b1f[2][2];

#define READ_A(BUF, SLOT)                                                                      \
  do {                                                                                         \
    if ((ABL & 1) && in_loop) break;                                                           \
    const char* _s = smem + (BUF) * BUF_BYTES + slot_offset(SLOT) + a_row_off;                 \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                         \
      af[mi][0] = *(const bf16x8*)(_s + mi * 2048 + fo0);                                      \
      af[mi][1] = *(const bf16x8*)(_s + mi * 2048 + fo1);                                      \
    }                                                                                          \
  } while (0)
#define READ_B(BUF, SLOT, DST)                                                                 \
  do {                                                                                         \
    if ((ABL & 1) && in_loop) break;                                                           \
    const char* _s = smem + (BUF) * BUF_BYTES + slot_offset(SLOT) + b_row_off;                 \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                         \
      DST[ni][0] = *(const bf16x8*)(_s + ni * 2048 + fo0);                                     \
      DST[ni][1] = *(const bf16x8*)(_s + ni * 2048 + fo1);                                     \
    }                                                                                          \
  } while (0)
#define MFMA_QA(QM, QN, BF, AF)                                                                \
  do {                                                                                         \
    if (ABL & 4) break;                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                             \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                           \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                         \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                       \
          acc[QM][QN][mi][ni] = VT_MFMA_16x16x32(BF[ni][kk], AF[mi][kk], acc[QM][QN][mi][ni]); \
    __builtin_amdgcn_s_setprio(0);                                                             \
  } while (0)
#define MFMA_Q(QM, QN, BF) MFMA_QA(QM, QN, BF, af)
  // one phase = load section | barrier | MFMA section + counted wait | barrier
#define SECTION_SPLIT()                        \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    __builtin_amdgcn_s_barrier();              \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)
  // end of a load section: the fragment reads must have LANDED before the barrier, so that the MFMA section that
  // follows starts issuing at once (the partner wave on this SIMD is in its MFMA section while we wait here)
#define LOAD_SECTION_END()                                   \
  do {                                                       \
    __builtin_amdgcn_sched_barrier(0);                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    SECTION_SPLIT();                                         \
  } while (0)
#define PHASE_END(G)                                              \
  do {                                                            \
    if ((G) < drain_from) VT_VMCNT(6); else VT_VMCNT(0);          \
    SECTION_SPLIT();                                              \
  } while (0)
  // 4-phase variant: COND = "the stage(s) this phase was due to issue were issued" (else nothing newer is in flight: drain)
#define PHASE_END4(COND, N)                                       \
  do {                                                            \
    if (COND) VT_VMCNT(N); else VT_VMCNT(0);                      \
    SECTION_SPLIT();                                              \
  } while (0)

  bool in_loop = false;
  const int nt = (u_end - u_begin) * 2;    // K steps of this workgroup (even, >= 4: checked by the launcher)
  const int total_phases = 4 * nt;
  const int drain_from = total_phases - 6;  // phases g >= this did not all issue a stage in (g-2..g): drain instead

  // consumer side of the norm fold: this lane's 8 row factors are requested ahead of the first stage and pinned in registers
  // behind the prologue's vmcnt(0) -- fetched at the top of the epilogue they cost every tile an exposed L2/HBM round trip
  // (+1.3 us per tile measured), and inside the row loop one per row
  float rsv[2][4];
  if constexpr (EPI != VT_EPI_F32_RESID && EPI != VT_EPI_F32) {
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        rsv[qm][mi] = p.nf.row_scale ? p.nf.row_scale[min(bm0 + qm * 128 + wr * 64 + mi * 16 + (lane & 15), p.M - 1)] : 1.f;
  }
  // ---- prologue: stages of global phases -6..-1 = K step 0 (A0 B0 B1 A1) + K step 1 (A0 B0) -----------------------------
  STAGE(0, SLOT_A0);
  STAGE(0, SLOT_B0);
  STAGE(0, SLOT_B1);
  STAGE(0, SLOT_A1);
  STAGE(1, SLOT_A0);
  STAGE(1, SLOT_B0);
  if (P4) STAGE(1, SLOT_B1);
  VT_VMCNT(0);
  if constexpr (EPI != VT_EPI_F32_RESID && EPI != VT_EPI_F32) {
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) asm volatile("" : "+v"(rsv[qm][mi]));   // the loads stay up here
  }
  SECTION_SPLIT();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // stagger: the second wave row runs one barrier behind
  __builtin_amdgcn_sched_barrier(0);

  // ablation builds need the fragments initialised once from valid LDS
  if (ABL & 1) { READ_B(0, SLOT_B0, b0f); READ_B(0, SLOT_B1, b1f); READ_A(0, SLOT_A0); }
  in_loop = true;
  // ---- 4-phase loop: one K step = phase A (reads A0 B0 B1, quadrants Q00 Q01) + phase B (reads A1, quadrants Q11 Q10) -------
  // Global phase g = 2u + {0, 1}. Stage schedule: phase 2u issues A1 of K step u+1; phase 2u+1 issues A0 B0 B1 of K step
  // u+2 into the slots K step u read one phase earlier (its A1 slot, read in this very phase, is only re-staged at phase
  // 2u+2). Every datum is staged THREE phases before its first read, as in the 8-phase schedule (six half-length phases).
  //   RAW  a read in load(p) needs every wave's DMA landed before the barrier that opens it: the reader's own group waited at
  //        M(p-1), the staggered group at M(p-2). M(even) = vmcnt(2) leaves only this phase's A1 stage in flight, M(odd) =
  //        vmcnt(6) only this phase's A0 B0 B1: everything staged at phases <= p-3 has landed at M(p-2). OK.
  //   WAR  a slot read in load(p) is re-staged at phase >= p+1 (A0 B0 B1: read 2u, staged 2u+1; A1: read 2u+1, staged 2u+2):
  //        the staggered group's reads of phase p finish (lgkmcnt(0)) one barrier before any wave issues phase p+1's stages. OK.
  if constexpr (P4) {
    for (int it = 0; it < nt / 2; ++it) {
      const int u = it * 2;
      // ===== K step u (buffer 0) =====
      if (u + 1 < nt) STAGE(1, SLOT_A1);
      READ_B(0, SLOT_B0, b0f);
      READ_B(0, SLOT_B1, b1f);
      READ_A(0, SLOT_A0);
      LOAD_SECTION_END();
      MFMA_Q(0, 0, b0f);
      MFMA_Q(0, 1, b1f);
      PHASE_END4(u + 1 < nt, 2);
      if (u + 2 < nt) {
        STAGE(0, SLOT_A0);
        STAGE(0, SLOT_B0);
        STAGE(0, SLOT_B1);
      }
      READ_A(0, SLOT_A1);
      LOAD_SECTION_END();
      MFMA_Q(1, 1, b1f);
      MFMA_Q(1, 0, b0f);
      PHASE_END4(u + 2 < nt, 6);
      // ===== K step u+1 (buffer 1) =====
      if (u + 2 < nt) STAGE(0, SLOT_A1);
      READ_B(1, SLOT_B0, b0f);
      READ_B(1, SLOT_B1, b1f);
      READ_A(1, SLOT_A0);
      LOAD_SECTION_END();
      MFMA_Q(0, 0, b0f);
      MFMA_Q(0, 1, b1f);
      PHASE_END4(u + 2 < nt, 2);
      if (u + 3 < nt) {
        STAGE(1, SLOT_A0);
        STAGE(1, SLOT_B0);
        STAGE(1, SLOT_B1);
      }
      READ_A(1, SLOT_A1);
      LOAD_SECTION_END();
      MFMA_Q(1, 1, b1f);
      MFMA_Q(1, 0, b0f);
      PHASE_END4(u + 3 < nt, 6);
    }
  } else
  for (int it = 0; it < nt / 2; ++it) {
    const int g0 = it * 8;                     // global phase of this iteration's first phase
    const int u = it * 2;                      // K step in buffer 0
    // ===== K step u (buffer 0) =====
    // phase 0: stage (u+1, B1) -> buffer 1; read A0,B0; Q00
    if (u + 1 < nt) STAGE(1, SLOT_B1);
    READ_B(0, SLOT_B0, b0f);
    READ_A(0, SLOT_A0);
    LOAD_SECTION_END();
    MFMA_Q(0, 0, b0f);
    PHASE_END(g0 + 0);
    // phase 1: stage (u+1, A1); read B1; Q01
    if (u + 1 < nt) STAGE(1, SLOT_A1);
    READ_B(0, SLOT_B1, b1f);
    LOAD_SECTION_END();
    MFMA_Q(0, 1, b1f);
    PHASE_END(g0 + 1);
    // phase 2: stage (u+2, A0) -> buffer 0; read A1; Q11
    if (u + 2 < nt) STAGE(0, SLOT_A0);
    READ_A(0, SLOT_A1);
    LOAD_SECTION_END();
    MFMA_Q(1, 1, b1f);
    PHASE_END(g0 + 2);
    // phase 3: stage (u+2, B0); no read; Q10
    if (u + 2 < nt) STAGE(0, SLOT_B0);
    LOAD_SECTION_END();
    MFMA_Q(1, 0, b0f);
    PHASE_END(g0 + 3);
    // ===== K step u+1 (buffer 1) =====
    if (u + 2 < nt) STAGE(0, SLOT_B1);
    READ_B(1, SLOT_B0, b0f);
    READ_A(1, SLOT_A0);
    LOAD_SECTION_END();
    MFMA_Q(0, 0, b0f);
    PHASE_END(g0 + 4);
    if (u + 2 < nt) STAGE(0, SLOT_A1);
    READ_B(1, SLOT_B1, b1f);
    LOAD_SECTION_END();
    MFMA_Q(0, 1, b1f);
    PHASE_END(g0 + 5);
    if (u + 3 < nt) STAGE(1, SLOT_A0);
    READ_A(1, SLOT_A1);
    LOAD_SECTION_END();
    MFMA_Q(1, 1, b1f);
    PHASE_END(g0 + 6);
    if (u + 3 < nt) STAGE(1, SLOT_B0);
    LOAD_SECTION_END();
    MFMA_Q(1, 0, b0f);
    PHASE_END(g0 + 7);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // pairs with the stagger barrier of the second wave row

  // ---- epilogue ------------------------------------------------------------------------------------------------------------
  // folded RMSNorm: consumer = accumulators of row m scaled by row_scale[m]; producer (residual epilogue) = also bf16(x_new *
  // out_w[n]) and the sum of x_new^2 per row and 32-column group (the 4 lanes lane>>4 of a row hold one group: 2 x 16 columns)
  const float* const row_scale = p.nf.row_scale;
  if constexpr (EPI == VT_EPI_QKV_PAGES) {
    // Fused QKV epilogue. A 256-column tile is two whole heads of ONE section (q, k or v: H % 256 == 0). Every lane writes its own
    // accumulator elements, as in the plain epilogue (a lane holds row m = ..+(lane & 15) and 4 consecutive dims per fragment):
    //   v    : the 4 lanes of a quad hold 4 consecutive rows (tokens) x 4 dims; a 4x4 transpose inside the quad (two xor-shuffles
    //          per fragment) turns that into 4 consecutive tokens of ONE dim per lane = one 8-byte store into the V^T page;
    //   q, k : the rotary partner of dim d is dim d +- 64, held by the wave two columns over: the eight waves park the head as
    //          bf16 in the LDS the main loop no longer needs -- bf16 because the reference rounds the projection to the storage
    //          dtype BEFORE the rotary embedding (the oracle's emulation does the same) -- every lane reads its partner values back
    //          and evaluates the same fp32 expression as vt_kv_tiles; q goes to the fused-QKV buffer, k to its page slot.
    // The last new row of a sequence also zero-fills the rest of its page (k rows, v^t columns), as vt_kv_tiles does.
    const int Hq = p.qf.heads * 128;
    const int sect = bn0 / Hq;                       // 0 = q, 1 = k, 2 = v (uniform over the workgroup)
    const int head0 = (bn0 - sect * Hq) >> 7;
    const int heads = p.qf.heads;
    int enc[8], rpos[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = min(bm0 + (r >> 2) * 128 + wr * 64 + (r & 3) * 16 + (lane & 15), p.M - 1);
      enc[r] = (sect >= 1) ? p.qf.row_slot[m] : 0;
      rpos[r] = (sect < 2) ? p.qf.positions[m] : 0;
    }
    if (sect == 2) {
#pragma unroll
      for (int qn = 0; qn < 2; ++qn) {
        const int head = head0 + qn;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int qm = r >> 2, mi = r & 3;
          const int m = bm0 + qm * 128 + wr * 64 + mi * 16 + (lane & 15);
          // slots of the quad's first and last row: 4 consecutive rows -> 4 consecutive slots of one page (the common case)
          const int e0 = __shfl(enc[r], lane & ~3, 64), e3 = __shfl(enc[r], lane | 3, 64);
          const bool quad_ok = (bm0 + qm * 128 + wr * 64 + mi * 16 + ((lane & 15) | 3)) < p.M &&
                               ((e3 & 0x3ffffff) - (e0 & 0x3ffffff)) == 3 && ((e0 & 3) == 0);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const f32x4 v = acc[qm][qn][mi][ni];
            const int n0 = wc * 32 + ni * 16 + ((lane >> 4) << 2);
            // 4x4 transpose over the quad: afterwards lane j = lane & 3 holds dim n0 + j of rows (lane & ~3) .. +3
            float t0 = v[0], t1 = v[1], t2 = v[2], t3 = v[3];
            {
              const float s01 = __shfl_xor((lane & 1) ? t0 : t1, 1, 64), s23 = __shfl_xor((lane & 1) ? t2 : t3, 1, 64);
              if (lane & 1) { t0 = s01; t2 = s23; } else { t1 = s01; t3 = s23; }
              const float u02 = __shfl_xor((lane & 2) ? t0 : t2, 2, 64), u13 = __shfl_xor((lane & 2) ? t1 : t3, 2, 64);
              if (lane & 2) { t0 = u02; t1 = u13; } else { t2 = u02; t3 = u13; }
            }
            // now t_i = element (row quad_base + i, dim n0 + (lane & 3))
            const int dim = n0 + (lane & 3);
            if (quad_ok) {
              const int sl = e0 & 0x3ffffff;
              u32x2 o;
              o.x = op2_to_f16x2(pack_op2(t0, t1));   // bf16 like the plain store epilogue, then the V^T pages' fp16 (vt_common.h)
              o.y = op2_to_f16x2(pack_op2(t2, t3));
              *(u32x2*)(p.qf.vt_pages + (((size_t)(sl >> 6) * heads + head) * 128 + dim) * 64 + (sl & 63)) = o;
            } else {
              const float tv[4] = {t0, t1, t2, t3};
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int ei = __shfl(enc[r], (lane & ~3) | i, 64);
                const int mi_row = bm0 + qm * 128 + wr * 64 + mi * 16 + ((lane & 12) | i);
                if (mi_row < p.M) {
                  const int sl = ei & 0x3ffffff;
                  p.qf.vt_pages[(((size_t)(sl >> 6) * heads + head) * 128 + dim) * 64 + (sl & 63)] = op_to_f16_bits(f32_to_op(tv[i]));
                }
              }
            }
            const int tl = (unsigned)enc[r] >> 26;      // this lane's OWN row ends a sequence: zero its 4 dims behind the slot
            if (tl && m < p.M) {
              const int sl = enc[r] & 0x3ffffff;
              for (int j = 0; j < 4; ++j) {
                bf16_t* col = p.qf.vt_pages + (((size_t)(sl >> 6) * heads + head) * 128 + n0 + j) * 64 + (sl & 63);
                for (int t = 1; t <= tl; ++t) col[t] = 0;
              }
            }
          }
        }
      }
    } else {
      bf16_t* const T = (bf16_t*)smem;
      constexpr int LDT = 136;                       // 256 rows x (128 + 8) bf16
      const float* __restrict__ const cos_t = p.qf.rope_cos;
      const float* __restrict__ const sin_t = p.qf.rope_sin;
      const bool upper = wc >= 2;                    // this wave holds dims 64..127 of the head: out = x*cos + partner*sin
#pragma unroll
      for (int qn = 0; qn < 2; ++qn) {
        const int head = head0 + qn;
        __syncthreads();                             // main loop / previous head: nobody reads this LDS any more
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int row = (r >> 2) * 128 + wr * 64 + (r & 3) * 16 + (lane & 15);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const f32x4 v = acc[r >> 2][qn][r & 3][ni];
            u32x2 o = {pack_op2(v[0], v[1]), pack_op2(v[2], v[3])};
            *(u32x2*)(T + row * LDT + wc * 32 + ni * 16 + ((lane >> 4) << 2)) = o;
          }
        }
        __syncthreads();
        // one row at a time: ALL its loads (partner values from LDS, cos / sin rows from L2) are issued before its first store --
        // a load behind a store to memory the compiler cannot tell apart waits for it, which serialises the L2 round trips
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
          u32x2 part[1][2];
          f32x4 c4[1][2], s4[1][2];
#pragma unroll
          for (int h = 0; h < 1; ++h) {
            const int r = rp + h;
            const int row = (r >> 2) * 128 + wr * 64 + (r & 3) * 16 + (lane & 15);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const int col = wc * 32 + ni * 16 + ((lane >> 4) << 2);
              part[h][ni] = *(const u32x2*)(T + row * LDT + (col ^ 64));
              c4[h][ni] = *(const f32x4*)(cos_t + (size_t)rpos[r] * 64 + (col & 63));
              s4[h][ni] = *(const f32x4*)(sin_t + (size_t)rpos[r] * 64 + (col & 63));
            }
          }
#pragma unroll
          for (int h = 0; h < 1; ++h) {
            const int r = rp + h;
            const int m = bm0 + (r >> 2) * 128 + wr * 64 + (r & 3) * 16 + (lane & 15);
            if (m >= p.M) continue;
            const int sl = enc[r] & 0x3ffffff, tl = (unsigned)enc[r] >> 26;
            bf16_t* __restrict__ dst = (sect == 0) ? (bf16_t*)p.C + (size_t)m * p.ldc + head * 128
                                                   : p.qf.k_pages + (((size_t)(sl >> 6) * heads + head) * 64 + (sl & 63)) * 128;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const int col = wc * 32 + ni * 16 + ((lane >> 4) << 2);
              const f32x4 v = acc[r >> 2][qn][r & 3][ni];
              const uint32_t xo[2] = {pack_op2(v[0], v[1]), pack_op2(v[2], v[3])};   // the bf16 values parked above
              const uint32_t po[2] = {part[h][ni].x, part[h][ni].y};
              uint32_t ov[2];
#pragma unroll
              for (int w = 0; w < 2; ++w) {
                // a = low-half value, b = high-half value of the pair; the shared rope_lo / rope_hi of vt_kv_tiles
                const float a0 = oplo_to_f32(upper ? po[w] : xo[w]), a1 = ophi_to_f32(upper ? po[w] : xo[w]);
                const float b0 = oplo_to_f32(upper ? xo[w] : po[w]), b1 = ophi_to_f32(upper ? xo[w] : po[w]);
                const float c0 = c4[h][ni][2 * w], c1 = c4[h][ni][2 * w + 1], s0 = s4[h][ni][2 * w], s1 = s4[h][ni][2 * w + 1];
                ov[w] = upper ? pack_op2(rope_hi(a0, b0, c0, s0), rope_hi(a1, b1, c1, s1))
                              : pack_op2(rope_lo(a0, b0, c0, s0), rope_lo(a1, b1, c1, s1));
              }
              const u32x2 o = {ov[0], ov[1]};
              *(u32x2*)(dst + col) = o;
              if (sect == 1 && tl) {                 // last new row of its sequence: zero the k rows behind it in the page
                const u32x2 z = {0u, 0u};
                for (int t = 1; t <= tl; ++t) *(u32x2*)(dst + t * 128 + col) = z;
              }
            }
          }
          if (rp & 1) __builtin_amdgcn_sched_barrier(0);   // at most two rows' loads in flight: no spills (accumulators stay live)
        }
      }
    }
  } else
  if constexpr (EPI == VT_EPI_F32_RESID) {
    // residual epilogue: C += acc (+ bias). The 4 x 16-byte reads of a row are issued together and one row ahead of the
    // stores (the compiler cannot move a load across the stores of the previous row: it cannot prove they do not alias), so
    // a wave has 8 read requests in flight instead of 1-4. With the norm fold the row's y values and partial sums follow.
    float* const __restrict__ Cf = (float*)p.C;
    float* const __restrict__ out_partials = p.nf.out_partials;
    bf16_t* const __restrict__ out_xw = p.nf.out_xw;
    const int lq = (lane >> 4) << 2;
    f32x4 w4[2][2], b4[2][2];
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = min(bn0 + qn * 128 + wc * 32 + ni * 16 + lq, p.N - 4);
        w4[qn][ni] = out_partials ? *(const f32x4*)(p.nf.out_w + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        b4[qn][ni] = p.bias ? *(const f32x4*)(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    auto row_of = [&](int r) { return bm0 + (r >> 2) * 128 + wr * 64 + (r & 3) * 16 + (lane & 15); };
    auto load_row = [&](int r, f32x4 (&dst)[2][2]) {
      const int m = min(row_of(r), p.M - 1);
#pragma unroll
      for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int n = min(bn0 + qn * 128 + wc * 32 + ni * 16 + lq, p.N - 4);
          dst[qn][ni] = *(const f32x4*)(Cf + (size_t)m * p.ldc + n);
        }
    };
    float rs8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) rs8[r] = row_scale ? row_scale[min(row_of(r), p.M - 1)] : 1.f;
    f32x4 cur[2][2], nxt[2][2];
    load_row(0, cur);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (r + 1 < 8) load_row(r + 1, nxt);
      const int m = row_of(r);
      const int qm = r >> 2, mi = r & 3;
      if (m < p.M) {
        const float rs = rs8[r];
#pragma unroll
        for (int qn = 0; qn < 2; ++qn) {
          const int nbase = bn0 + qn * 128 + wc * 32;
          float ss = 0.f;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const int n = nbase + ni * 16 + lq;
            if (n >= p.N) continue;
            const f32x4 nv = cur[qn][ni] + (acc[qm][qn][mi][ni] * rs + b4[qn][ni]);
            *(f32x4*)(Cf + (size_t)m * p.ldc + n) = nv;
            if (out_partials) {
              u32x2 o;
              o.x = pack_op2(nv[0] * w4[qn][ni][0], nv[1] * w4[qn][ni][1]);
              o.y = pack_op2(nv[2] * w4[qn][ni][2], nv[3] * w4[qn][ni][3]);
              *(u32x2*)(out_xw + (size_t)m * p.nf.ld_xw + n) = o;
              ss += (nv[0] * nv[0] + nv[1] * nv[1]) + (nv[2] * nv[2] + nv[3] * nv[3]);
            }
          }
          if (out_partials && nbase < p.N) {   // N % 32 == 0: the group is whole or absent, uniform over the row's 4 lanes
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            if ((lane >> 4) == 0) out_partials[(size_t)(nbase >> 5) * p.nf.out_ldp + m] = ss;
          }
        }
      }
#pragma unroll
      for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) cur[qn][ni] = nxt[qn][ni];
    }
  } else {
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = bm0 + qm * 128 + wr * 64 + mi * 16 + (lane & 15);
        if (m >= p.M) continue;
        const float rs = (EPI == VT_EPI_F32) ? 1.f : rsv[qm][mi];
#pragma unroll
        for (int qn = 0; qn < 2; ++qn) {
          const int nbase = bn0 + qn * 128 + wc * 32;     // multiple of 32
          if constexpr (EPI == VT_EPI_SWIGLU_BF16) {
            if (nbase + ((lane >> 4) << 2) >= p.N) continue;
            const f32x4 g = acc[qm][qn][mi][0] * rs, u2 = acc[qm][qn][mi][1] * rs;
            u32x2 o;
            o.x = pack_op2(silu8(g[0]) * u2[0], silu8(g[1]) * u2[1]);
            o.y = pack_op2(silu8(g[2]) * u2[2], silu8(g[3]) * u2[3]);
            *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + (nbase >> 1) + ((lane >> 4) << 2)) = o;
          } else {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const int n = nbase + ni * 16 + ((lane >> 4) << 2);
              if (n >= p.N) continue;
              f32x4 v = acc[qm][qn][mi][ni] * rs;
              if (p.bias && split == 0) v += *(const f32x4*)(p.bias + n);   // split-K: the bias goes in once
              if constexpr (EPI == VT_EPI_BF16_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf8(v[r]);
              } else if constexpr (EPI == VT_EPI_BF16_QGELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = quick_gelu8(v[r]);
              } else if constexpr (EPI == VT_EPI_BF16_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
              }
              if constexpr (EPI == VT_EPI_F32) {
                *(f32x4*)((float*)p.C + (size_t)split * p.slab + (size_t)m * p.ldc + n) = v;
              } else {
                u32x2 o;
                o.x = pack_op2(v[0], v[1]);
                o.y = pack_op2(v[2], v[3]);
                *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
              }
            }
          }
        }
      }
  }
}

// Epilogue of the four-wave kernel: lane holds row m = .. + (lane & 15) and 4 consecutive columns of every 16-column fragment.
// One wave per SIMD: nothing overlaps a round trip to memory here. The fp32 residual rows are therefore requested THREE fragment
// rows ahead of their use (the fragment registers of the main loop are free by now; rows / columns past the edge are clamped for
// the loads and masked for the stores, the bias is fetched once): with a load / wait / store round per fragment row the
// read-modify-write of C cost 12 % of a one-round o_proj launch (5120 x 4096 x 4096: 1245 -> 1292 TFLOP/s, 1381 with a bf16 store).
// The store-only epilogues keep the plain skip-past-the-edge form (the same restructuring measured 8 % SLOWER on them at K = 1024).
// LDS-staged store epilogue (bf16 outputs of the 256 / 320-row tiles): the direct form above it writes, per store instruction, 8 bytes
// per lane = 32 contiguous bytes in each of 16 rows -- 64 instructions per wave whose cost is per INSTRUCTION and per touched line, not
// per byte (tools/vit_gemm_diag.py: 28 us of a 46 us K = 1024 launch do not depend on K). Here every wave parks its 128-column rows
// as bf16 in the LDS the main loop no longer needs (wave-private region, 16-byte chunks XOR-swizzled by the row: conflict-free for
// the 8-byte fragment writes and for the 16-byte row reads), reads them back row-contiguous and stores 16 bytes per lane: one
// instruction = 4 rows x 256 contiguous bytes (whole 128-byte lines), half as many instructions.
template <int EPI, int MT>
__device__ __forceinline__ void w4_epilogue_lds(const GemmP8& p, f32x4 (&acc)[MT][8], int bm0, int bn0, int wr, int wc, int lane, char* smem,
                                                int wave) {
  static_assert(EPI == VT_EPI_BF16 || EPI == VT_EPI_BF16_GELU || EPI == VT_EPI_BF16_QGELU || EPI == VT_EPI_BF16_RELU || EPI == VT_EPI_SWIGLU_BF16, "bf16 store epilogues");
  constexpr bool SW = EPI == VT_EPI_SWIGLU_BF16;
  constexpr int RC = (MT <= 8) ? MT : 5;             // fragment rows per pass (MT = 10: two passes of 80 rows)
  constexpr int ROWB = SW ? 128 : 256;               // bytes per parked row (64 | 128 bf16 columns)
  constexpr int CH = ROWB / 16;                      // 16-byte chunks per row
  constexpr int RPI = 64 / CH;                       // rows per read / store instruction
  char* const my = smem + wave * (RC * 16 * ROWB);
  const int r16 = lane & 15, cg = lane >> 4;
  bf16_t* const C = (bf16_t*)p.C;
  const int ncol0 = SW ? ((bn0 >> 1) + wc * 64) : (bn0 + wc * 128);   // first output column of this wave
  const int nout = SW ? (p.N >> 1) : p.N;
  f32x4 b4[8];
  if constexpr (!SW) {
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
      b4[ni] = p.bias ? *(const f32x4*)(p.bias + min(bn0 + wc * 128 + ni * 16 + (cg << 2), p.N - 4)) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  __builtin_amdgcn_s_barrier();                      // every wave is done with the main loop's LDS image
#pragma unroll
  for (int c0 = 0; c0 < MT; c0 += RC) {
#pragma unroll
    for (int mi = c0; mi < c0 + RC; ++mi) {
      const int row = (mi - c0) * 16 + r16;
      if constexpr (SW) {
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
          const f32x4 g = acc[mi][2 * nj], u2 = acc[mi][2 * nj + 1];
          u32x2 o;
          o.x = pack_op2(silu8(g[0]) * u2[0], silu8(g[1]) * u2[1]);
          o.y = pack_op2(silu8(g[2]) * u2[2], silu8(g[3]) * u2[3]);
          *(u32x2*)(my + row * ROWB + (((nj * 2 + (cg >> 1)) ^ (row & (CH - 1))) << 4) + (cg & 1) * 8) = o;
        }
      } else {
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
          f32x4 v = acc[mi][ni] + b4[ni];
          if constexpr (EPI == VT_EPI_BF16_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf8(v[r]);
          } else if constexpr (EPI == VT_EPI_BF16_QGELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = quick_gelu8(v[r]);
          } else if constexpr (EPI == VT_EPI_BF16_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          u32x2 o;
          o.x = pack_op2(v[0], v[1]);
          o.y = pack_op2(v[2], v[3]);
          *(u32x2*)(my + row * ROWB + (((ni * 2 + (cg >> 1)) ^ (row & (CH - 1))) << 4) + (cg & 1) * 8) = o;
        }
      }
    }
    // LDS operations of one wave complete in order: the reads below see the writes above without a block barrier -- but the COMPILER must
    // not move them across each other (it may take the 8-byte stores and the 16-byte loads for non-aliasing)
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const int m_base = bm0 + wr * (MT * 16) + c0 * 16;
#pragma unroll
    for (int j = 0; j < RC * 16 / RPI; ++j) {
      const int row = j * RPI + lane / CH, ck = lane % CH;
      const u32x4 d = *(const u32x4*)(my + row * ROWB + ((ck ^ (row & (CH - 1))) << 4));
      const int m = m_base + row, n = ncol0 + ck * 8;
      if (m < p.M && n < nout) *(u32x4*)(C + (size_t)m * p.ldc + n) = d;
    }
    __builtin_amdgcn_wave_barrier();                 // (the next pass overwrites the region)
    asm volatile("" ::: "memory");
  }
}

template <int EPI, int MT, int NI = 8>
__device__ __forceinline__ void w4_epilogue(const GemmP8& p, f32x4 (&acc)[MT][NI], int bm0, int bn0, int wr, int wc, int lane, int split = 0) {
  constexpr int WN = NI * 16;   // columns per wave
  if constexpr (EPI == VT_EPI_F32_RESID) {
    const int row0 = bm0 + wr * (MT * 16) + (lane & 15);
    const int col0 = bn0 + wc * WN + ((lane >> 4) << 2);
    const bool has_bias = p.bias != nullptr;
    f32x4 b4[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      b4[ni] = has_bias ? *(const f32x4*)(p.bias + min(col0 + ni * 16, p.N - 4)) : (f32x4){0.f, 0.f, 0.f, 0.f};
    float* const Cf = (float*)p.C;
    constexpr int PF = 3;
    f32x4 cur[PF][NI];
#define W4E_LOAD(MI)                                                                                              \
  do {                                                                                                            \
    const float* _r = Cf + (size_t)min(row0 + (MI) * 16, p.M - 1) * p.ldc;                                        \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) cur[(MI) % PF][ni] = *(const f32x4*)(_r + min(col0 + ni * 16, p.N - 4)); \
  } while (0)
#pragma unroll
    for (int mi = 0; mi < PF; ++mi) W4E_LOAD(mi);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int m = row0 + mi * 16;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = col0 + ni * 16;
        const f32x4 v = has_bias ? acc[mi][ni] + b4[ni] : acc[mi][ni];
        const f32x4 o = cur[mi % PF][ni] + v;
        if (m < p.M && n < p.N) *(f32x4*)(Cf + (size_t)m * p.ldc + n) = o;
      }
      if (mi + PF < MT) W4E_LOAD(mi + PF);
    }
#undef W4E_LOAD
  } else {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int m = bm0 + wr * (MT * 16) + mi * 16 + (lane & 15);
      if (m >= p.M) continue;
      if constexpr (EPI == VT_EPI_SWIGLU_BF16) {
#pragma unroll
        for (int nj = 0; nj < NI / 2; ++nj) {
          const int nbase = bn0 + wc * WN + nj * 32;
          if (nbase + ((lane >> 4) << 2) >= p.N) continue;
          const f32x4 g = acc[mi][2 * nj], u2 = acc[mi][2 * nj + 1];
          u32x2 o;
          o.x = pack_op2(silu8(g[0]) * u2[0], silu8(g[1]) * u2[1]);
          o.y = pack_op2(silu8(g[2]) * u2[2], silu8(g[3]) * u2[3]);
          *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + (nbase >> 1) + ((lane >> 4) << 2)) = o;
        }
      } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int n = bn0 + wc * WN + ni * 16 + ((lane >> 4) << 2);
          if (n >= p.N) continue;
          f32x4 v = acc[mi][ni];
          if (p.bias && split == 0) v += *(const f32x4*)(p.bias + n);   // split-K: the bias goes in once
          if constexpr (EPI == VT_EPI_BF16_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf8(v[r]);
          } else if constexpr (EPI == VT_EPI_BF16_QGELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = quick_gelu8(v[r]);
          } else if constexpr (EPI == VT_EPI_BF16_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          if constexpr (EPI == VT_EPI_F32) {
            *(f32x4*)((float*)p.C + (size_t)split * p.slab + (size_t)m * p.ldc + n) = v;
          } else {
            u32x2 o;
            o.x = pack_op2(v[0], v[1]);
            o.y = pack_op2(v[2], v[3]);
            *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// gemm_w4_kernel: 256x256x64 tile, FOUR waves (2 x 2), 128x128 per wave, ONE wave per SIMD (512-register budget: the 64 16x16
// accumulators = 256 registers pinned to the accumulator file, two sets of 8 + 8 fragment quads = 128 VGPRs).
//
// Why: per K step a wave of this shape reads 32 KB of fragments from LDS for 4.2 MFLOP (15 KB/MFLOP); the 8-wave ping-pong tile
// (64x32 quadrants) reads 24 KB for 2.1 MFLOP (22.9 KB/MFLOP) and leaves a third of the matrix pipe idle in hand-offs between its
// two wave rows (PMC: 67 % MFMA busy). With one instruction stream per SIMD nothing hides a memory instruction behind
// another wave, so the K step is scheduled by hand, MFMA by MFMA (the MFMAs are asm statements - pinned accumulators, opaque to the
// scheduler - and every gap is fenced with sched_barrier). What the timing ablations of the first versions showed (DESIGN.md 3.1:
// the 16 LDS-DMA pieces cost 15 %, the 32 fragment reads 8 %, a barrier nothing) shaped the schedule:
//   * at most ONE memory instruction per MFMA gap, every second gap;
//   * LDS-DMA through a buffer resource (`buffer_load_dwordx4 .. offen lds`: SGPR base + one constant VGPR offset per piece + the
//     K offset in an SGPR) - no address arithmetic on the vector pipe;
//   * the A and B halves of an LDS buffer are released and refilled SEPARATELY (four barriers per K step), which spreads the 16
//     pieces over three quarters of the K step and leaves every piece 1.25 K steps to land (counted vmcnt).
//   K step t lives in LDS buffer b = t & 1 (same half-tile slots and source-side XOR swizzle as the ping-pong kernel; wave (wr, wc)
//   reads half tile A{wr} and B{wc}); MFMAs 0..63 work on fragment set 0 (k-half 0), 64..127 on set 1; a set is refilled only
//   while the other one is in use. Gap n = the slot behind MFMA n:
//       gap   0..14   read B fragments of k-half 1 (buffer b)             19: lgkmcnt(0)   20: barrier  -> B half of buffer b is free
//       gap  21..51   DMA B of K step t+2 -> buffer b | read A fragments of k-half 1
//                                                                          55: lgkmcnt(0)   56: barrier  -> A half of buffer b is free
//       gap  57..61   DMA A of K step t+2 (3 pieces)                      62: vmcnt(19)    63: barrier  -> B of K step t+1 has landed
//       gap  64..81   read B fragments of (t+1, k-half 0) from buffer b^1 | DMA A (5 pieces)
//                                                                          92: vmcnt(16)    93: barrier  -> A of K step t+1 has landed
//       gap  94..108  read A fragments of (t+1, k-half 0)
//   vmcnt: a wave issues its pieces in the order B(t) A(t) B(t+1) A(t+1) ..; at gap 62 the ones younger than B(t+1) are A(t+1) 8 +
//   B(t+2) 8 + A(t+2) 3, at gap 92 the ones younger than A(t+1) are B(t+2) 8 + A(t+2) 8.
//   WAR: the last reads of a half (lgkmcnt(0)) precede the barrier behind which ANY wave refills it. RAW: every wave waits for its
//   own pieces of a half before the barrier behind which any wave reads it.
// ------------------------------------------------------------------------------------------------------------------
// MT = 16-row fragments per wave along M: 8 -> the 256x256 tile above; 10 -> a 320x256 tile (160x128 per wave: 80 accumulator quads,
// the last 16 of them in VGPRs; 144 KiB of LDS; 160 MFMAs, 18 pieces and 36 fragment reads per K step, same gap pattern stretched):
// for row counts that fill WHOLE rounds of the 256 CUs only with 320-row tiles (5120 x 4096: 256 tiles instead of 320 = 1.25 rounds).
// ABL (timing ablations, wrong results; test library only): 1 = no LDS-DMA in the loop, 2 = no barriers, 4 = no fragment reads, 8 = no
// counted waits, 16 = unswizzled DMA sources. AUX_A / AUX_B = cache-policy bits of the DMA instructions (1 sc0, 2 nt, 16 sc1): measured,
// sc0 / sc1 make no difference and nt costs 20 %, so they stay 0.
#ifdef VT_ABLATIONS   // exploration knobs of the tile -> workgroup mapping (test library; VT_W4_GROUP_M / VT_W4R_GROUP_M / VT_W4X_GROUP_M / VT_W4_NOREMAP, read at the first launch)
__device__ int g_w4_knobs[4];
static void w4_knobs_once() {
  static bool done = false;
  if (done) return;
  done = true;
  int h[4] = {getenv("VT_W4_GROUP_M") ? atoi(getenv("VT_W4_GROUP_M")) : 0, getenv("VT_W4R_GROUP_M") ? atoi(getenv("VT_W4R_GROUP_M")) : 0,
              getenv("VT_W4_NOREMAP") ? atoi(getenv("VT_W4_NOREMAP")) : 0, getenv("VT_W4X_GROUP_M") ? atoi(getenv("VT_W4X_GROUP_M")) : 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_w4_knobs), h, sizeof(h));
}
#define W4_KNOB(i, dflt) (g_w4_knobs[i] > 0 ? g_w4_knobs[i] : (dflt))
#else
#define W4_KNOB(i, dflt) (dflt)
static inline void w4_knobs_once() {}
#endif
// The wait states between the LAST MFMAs of a main loop (asm statements: the compiler does not know their latency) and the first read of
// their results. They sit INSIDE the loop, on its last trip only: a statement behind the loop orders nothing against the register copies
// the compiler places on the loop's exit edge (the accumulators have a second definition on the zero-trip path, and reconciling the two
// assignments reads them right behind the loop branch). Round 6: with another register assignment those copies took a[252:255] -- the
// destination of the very last MFMA -- four instructions after its issue, and the SwiGLU epilogue of the 256-row tile stored stale values
// (tests/test_gpu_fp16.py::test_fp16_gemm_tile_kernels[300-384-256-13]); the bare asm volatile("s_nop ..") behind the loop had only
// ever worked because the copies happened to start with older accumulators. tools/check_scratch_window.py checks the generated code.
#define VT_MFMA_TAIL_WAIT(LAST_TRIP) do { if (LAST_TRIP) asm volatile("s_nop 15\n\ts_nop 7"); } while (0)
__host__ __device__ __forceinline__ constexpr int w4_vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }   // s_waitcnt vmcnt(n) only

template <int EPI, int MT = 8, int ABL = 0, int AUX_A = 0, int AUX_B = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4_kernel(GemmP8 p) {
  static_assert(MT == 7 || MT == 8 || MT == 10, "wave tile height: 112, 128 or 160 rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = MT * 32;                       // tile rows
  constexpr int A_BYTES = BM * 128, B_BYTES = 256 * 128, BUFB = A_BYTES + B_BYTES;   // one K step in LDS: A rows, then B rows
  constexpr int NPA = MT, NPB = 8, NP = NPA + NPB;  // LDS-DMA pieces (8 rows x 128 B) per wave and K step
  constexpr int H = MT * 8, NM = 2 * H;             // MFMAs per k-half / per K step
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + 255) / 256;
  const int nwg = tiles_m * tiles_n;
  // two-pass split-K (EPI == F32 only, as in the ping-pong kernel): split s of every tile writes its partial product to slab s
  const int ksplit = (EPI == VT_EPI_F32) ? p.ksplit : 1;
  const int split = (ksplit > 1) ? (int)blockIdx.x / nwg : 0;
  // Workgroup id -> place in the logical tile order. Blocks are dispatched round-robin over the 8 XCDs; xcd_remap gives every XCD ONE contiguous
  // eighth of the order (worth 2.5 % of the C3 step against no remap). A grid of WHOLE rounds (a multiple of 256 tiles, more than one round) is
  // walked round by round instead -- XCD x takes places [256 r + 32 x, + 32) of round r, so that the eight XCDs work on adjacent 4 x 8 tile blocks
  // at any time (round 6, bit-identical; tools/remap_ab.py: qkv 40960 x 12288 x 4096 2901 -> 2757 us, 10240 rows 715 -> 697, 5120 rows 0.5-2.7 % by
  // box, o_proj at 40960 rows 1112 -> 1092; six paired C3 steps -0.21 ms). With a partial last round (gate/up: 1720 tiles) it measured 0.5 % slower.
  const int b_ = (int)blockIdx.x - split * nwg;
  const bool whole_rounds = (nwg & 255) == 0 && nwg > 256 && W4_KNOB(2, 0) != 3;
  const int sid = W4_KNOB(2, 0) == 1 ? b_ : whole_rounds ? ((b_ >> 8) << 8) + (b_ & 7) * 32 + ((b_ >> 3) & 31) : xcd_remap(b_, nwg);
  const int ku = p.K >> 7;   // K in units of 128 (two K steps), balanced over the splits
  const int u_begin = (int)((long)ku * split / ksplit), u_end = (int)((long)ku * (split + 1) / ksplit);
  // tile groups of 4 rows x all columns once there are 8 or more row blocks (round 6, bit-identical outputs: gate/up 5120 x 22016 x 4096
  // 657 -> 640 us over three interleaved runs, the C3 step -0.45 ms; groups of 2 / 3 the same, 1 and 16 / 20 slower; with 3-5 row blocks --
  // the single-image shapes -- a group of 4 leaves a ragged last group and measured +0.4 %): an XCD's 32 concurrent tiles are then
  // 4 row blocks x 8 column tiles instead of 8 x 4
  const int GROUP_M = W4_KNOB(0, tiles_m >= 8 ? 4 : 8);
  const int per_group = GROUP_M * tiles_n;
  const int first_m = (sid / per_group) * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (sid % per_group) % gsz;
  const int tn = (sid % per_group) / gsz;
  const int bm0 = tm * BM, bn0 = tn * 256;

  // DMA sources. Pieces 0..7 of a K step are this wave's share of B (rows (8 wave + j) * 8 ..), pieces 8.. its share of A.
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)bm0 * p.lda + u_begin * 128), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)bn0 * p.ldw + u_begin * 128), 0, 0x7fffffff, 0x00020000);
  const int lrow = lane >> 3, lchk = lane & 7;
  int voff[18];   // NP entries used; sized by a literal: with a size that depends on MT, clang's HOST pass silently drops the kernel stub
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = (i < NPB ? wave * NPB + i : wave * NPA + (i - NPB)) * 8 + lrow;   // row inside the tile
    const int coff = ((ABL & 16) ? lchk : (lchk ^ ((row >> 1) & 7))) * 16;            // source-side XOR swizzle, bytes
    voff[i] = i < NPB ? (min(bn0 + row, p.N - 1) - bn0) * p.ldw * 2 + coff : (min(bm0 + row, p.M - 1) - bm0) * p.lda * 2 + coff;
  }
#define W4S_DMA_X(BUF, P, KB, AUX)                                                                                    \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                                      \
      (P) < NPB ? rsrc_b : rsrc_a,                                                                               \
      (__attribute__((address_space(3))) void*)(smem + (BUF) * BUFB +                                            \
                                                 ((P) < NPB ? A_BYTES + (wave * NPB + (P)) * 1024 : (wave * NPA + (P) - NPB) * 1024)), \
      16, voff[P], KB, 0, AUX)
#define W4S_DMA(BUF, P, KB) do { if ((P) < NPB) W4S_DMA_X(BUF, P, KB, AUX_B); else W4S_DMA_X(BUF, P, KB, AUX_A); } while (0)

  const int f = (lane >> 1) & 7;
  const int fo[2] = {(lane & 15) * 128 + (((lane >> 4) ^ f) << 4), (lane & 15) * 128 + ((((lane >> 4) | 4) ^ f) << 4)};
  const int a_base = wr * (MT * 16 * 128);
  const int b_base = A_BYTES + wc * (128 * 128);

  f32x4 acc[MT][8];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][MT], fb[2][8];
  // MFMA n of a K step: fragment set n / H, accumulator row (n % H) / 8, column n % 8; rows 0..7 live in the accumulator file
#define W4S_MFMA(N)                                                                                              \
  do {                                                                                                           \
    if ((((N) % H) >> 3) < 8)                                                                                    \
      asm volatile(VT_MFMA_16x16x32_ASM " %0, %1, %2, %0"                                                     \
                   : "+a"(acc[(((N) % H) >> 3) & 7][(N) & 7]) : "v"(fb[((N) / H) & 1][(N) & 7]), "v"(fa[((N) / H) & 1][(((N) % H) >> 3) % MT])); \
    else                                                                                                         \
      asm volatile(VT_MFMA_16x16x32_ASM " %0, %1, %2, %0"                                                     \
                   : "+v"(acc[(((N) % H) >> 3) % MT][(N) & 7]) : "v"(fb[((N) / H) & 1][(N) & 7]), "v"(fa[((N) / H) & 1][(((N) % H) >> 3) % MT])); \
  } while (0)
#define W4S_RDB(SET, BUF, KK, I) fb[SET][I] = *(const bf16x8*)(smem + (BUF) * BUFB + b_base + fo[KK] + (I) * 2048)
#define W4S_RDA(SET, BUF, KK, I) fa[SET][I] = *(const bf16x8*)(smem + (BUF) * BUFB + a_base + fo[KK] + (I) * 2048)
#define W4S_LGKM0() __builtin_amdgcn_s_waitcnt(0xc07f)

  const int nt = (u_end - u_begin) * 2;            // K steps of this workgroup's K range (K % 128 == 0, checked by the launcher)
  const int kb1 = min(1, nt - 1) * 128;
#pragma unroll
  for (int i = 0; i < NP; ++i) W4S_DMA(0, i, 0);
#pragma unroll
  for (int i = 0; i < NP; ++i) W4S_DMA(1, i, kb1);
  __builtin_amdgcn_s_waitcnt(w4_vmcnt_imm(NP));    // K step 0 has landed, K step 1 may still be in flight
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) W4S_RDB(0, 0, 0, i);
#pragma unroll
  for (int i = 0; i < MT; ++i) W4S_RDA(0, 0, 0, i);
  W4S_LGKM0();
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 4");                         // accumulator zeroes (v_accvgpr_write) -> first MFMA reading them as C

  // The memory / synchronisation instruction that follows MFMA n of a K step living in LDS buffer B (gap table of the header; the
  // positions for MT = 10 are the same pattern stretched: 18 slots in the first section, 8 early + 2 late A pieces).
  constexpr int S1_N = NPB + MT;                    // slots of section 1: DMA B / read A alternating, then the extra A reads
  constexpr int LG2 = 21 + 2 * S1_N + (MT == 7 ? 0 : 2), BAR2 = LG2 + 1;    // 55, 56  | 59, 60  | MT = 7: 51, 52 (the k-half is 56 MFMAs)
  constexpr int NE = MT == 7 ? 1 : MT == 8 ? 3 : 8; // A pieces issued before the k-half boundary
  constexpr int VM3 = BAR2 + 2 * NE, BAR3 = VM3 + 1;                        // 62, 63  | 76, 77  | 54, 55
  constexpr int VM4 = MT == 7 ? 84 : MT == 8 ? 92 : 100, BAR4 = VM4 + 1;    // 92, 93  | 100, 101 | 84, 85
  static_assert(BAR3 < H && BAR4 + 2 * MT < NM && H + 4 * (NPA - NE - 1) + 1 < VM4, "gap table");
#define W4S_IX(x, m) ((((x) % (m)) + (m)) % (m))   /* keeps the indices of the untaken branches inside their arrays */
#define W4S_GAP(n, B, KB)                                                                                        \
  do {                                                                                                           \
    if ((n) < 16) { if (!((n) & 1) && !(ABL & 4)) W4S_RDB(1, B, 1, ((n) >> 1) & 7); }                            \
    else if ((n) == 19 || (n) == LG2) { if (!(ABL & 8)) W4S_LGKM0(); }                                           \
    else if ((n) == 20 || (n) == BAR2 || (n) == BAR3 || (n) == BAR4) { if (!(ABL & 2)) __builtin_amdgcn_s_barrier(); } \
    else if ((n) >= 21 && (n) < 21 + 2 * S1_N) {                                                                 \
      if ((n) & 1) {                                                                                             \
        if ((((n) - 21) >> 1) >= 16) { if (!(ABL & 4)) W4S_RDA(1, B, 1, W4S_IX((((n) - 21) >> 1) - 8, MT)); }         \
        else if (((n) & 3) == 1) { if (!(ABL & 1)) W4S_DMA(B, W4S_IX(((n) - 21) >> 2, 8), KB); }                      \
        else { if (!(ABL & 4) && (((n) - 23) >> 2) < MT) W4S_RDA(1, B, 1, W4S_IX(((n) - 23) >> 2, MT)); }             \
      }                                                                                                          \
    }                                                                                                            \
    else if ((n) > BAR2 && (n) < VM3) { if (((n) - BAR2) & 1) { if (!(ABL & 1)) W4S_DMA(B, NPB + W4S_IX(((n) - BAR2) >> 1, NE), KB); } } \
    else if ((n) == VM3) { if (!(ABL & 9)) __builtin_amdgcn_s_waitcnt(w4_vmcnt_imm(NPA + NPB + NE)); }           \
    else if ((n) >= H && (n) < VM4) {                                                                            \
      if (!(((n) - H) & 1)) { if ((n) <= H + 14 && !(ABL & 4)) W4S_RDB(0, (B) ^ 1, 0, W4S_IX(((n) - H) >> 1, 8)); }   \
      else if ((((n) - H) & 3) == 1 && (((n) - H) >> 2) < NPA - NE) { if (!(ABL & 1)) W4S_DMA(B, NPB + NE + W4S_IX(((n) - H) >> 2, NPA - NE), KB); } \
    }                                                                                                            \
    else if ((n) == VM4) { if (!(ABL & 9)) __builtin_amdgcn_s_waitcnt(w4_vmcnt_imm(NPA + NPB)); }                \
    else if ((n) > BAR4 && (n) <= BAR4 + 2 * MT && (((n) - BAR4) & 1)) { if (!(ABL & 4)) W4S_RDA(0, (B) ^ 1, 0, W4S_IX(((n) - BAR4) >> 1, MT)); } \
  } while (0)
#define W4S_N(n, B, KB) W4S_MFMA(n); W4S_GAP(n, B, KB); __builtin_amdgcn_sched_barrier(0);
#define W4S_8(n, B, KB) W4S_N((n), B, KB) W4S_N((n) + 1, B, KB) W4S_N((n) + 2, B, KB) W4S_N((n) + 3, B, KB) \
                        W4S_N((n) + 4, B, KB) W4S_N((n) + 5, B, KB) W4S_N((n) + 6, B, KB) W4S_N((n) + 7, B, KB)
#define W4S_32(n, B, KB) W4S_8((n), B, KB) W4S_8((n) + 8, B, KB) W4S_8((n) + 16, B, KB) W4S_8((n) + 24, B, KB)
#define W4S_KSTEP(B, T)   /* spelled out: the optimiser does not unroll a loop around asm statements */           \
  do {                                                                                                           \
    const int _kb = min((T) + 2, nt - 1) * 128;    /* past the end of K: re-fetch the last K step, harmless */    \
    W4S_32(0, B, _kb) W4S_32(32, B, _kb) W4S_32(64, B, _kb) W4S_8(96, B, _kb) W4S_8(104, B, _kb)                  \
    if constexpr (NM > 112) { W4S_8(112, B, _kb) W4S_8(120, B, _kb) }                                            \
    if constexpr (NM > 128) { W4S_32(128, B, _kb) }                                                              \
  } while (0)
  for (int t = 0; t < nt; t += 2) {
    W4S_KSTEP(0, t);
    W4S_KSTEP(1, t + 1);
    VT_MFMA_TAIL_WAIT(t + 2 >= nt);
  }
  VT_VMCNT(0);                                     // the tail's redundant pieces
  asm volatile("s_nop 15\n\ts_nop 7");            // last MFMA result -> the epilogue's v_accvgpr_read (the compiler cannot see the hazard)
#undef W4S_KSTEP
#undef W4S_32
#undef W4S_8
#undef W4S_N
#undef W4S_GAP
#undef W4S_IX
#undef W4S_LGKM0
#undef W4S_RDA
#undef W4S_RDB
#undef W4S_MFMA
#undef W4S_DMA
#undef W4S_DMA_X
  constexpr bool bf16_store = EPI == VT_EPI_BF16 || EPI == VT_EPI_BF16_GELU || EPI == VT_EPI_BF16_QGELU || EPI == VT_EPI_BF16_RELU || EPI == VT_EPI_SWIGLU_BF16;
  if constexpr (ABL & 32) {                          // timing ablation: no epilogue stores at all
    float t = 0.f;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) t += acc[mi][ni][0] + acc[mi][ni][3];
    if (t == 1.2345e-30f) *(float*)p.C = t;
  } else {
    // The epilogue takes the lane id through an opaque copy: what it derives from it (row / column offsets, masks) is computed HERE and not hoisted in
    // front of the main loop, where -- 256 + 256 registers being taken -- it was spilled: round 6 found a scratch store between the prologue's
    // DMA pieces and their counted vmcnt wait in <16-bit store, 256-row tile> of the bf16 build (C3-224's qkv). A scratch access counts in
    // vmcnt and may retire out of order with the loads, so the counted wait could let a piece through early. tests/test_host_logic.py now
    // checks the generated code of every instantiation for scratch accesses between the first DMA and the last MFMA.
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    if constexpr (bf16_store && !(ABL & 64)) {
      w4_epilogue_lds<EPI, MT>(p, acc, bm0, bn0, wr, wc, lane_e, smem, wave);    // (launch_w4 checked alignment and N % 8)
    } else {
      w4_epilogue<EPI, MT>(p, acc, bm0, bn0, wr, wc, lane_e, split);
    }
  }
}

template <int EPI, int MT = 8, int ABL = 0, int AUX_A = 0, int AUX_B = 0>
int launch_w4(const GemmP8& p, hipStream_t s) {
  constexpr int BM = MT * 32;
  constexpr int smem = 2 * (BM * 128 + 256 * 128);  // 128 KiB | 144 KiB
  constexpr bool bf16_store = EPI == VT_EPI_BF16 || EPI == VT_EPI_BF16_GELU || EPI == VT_EPI_BF16_QGELU || EPI == VT_EPI_BF16_RELU || EPI == VT_EPI_SWIGLU_BF16;
  if constexpr (bf16_store && !(ABL & (32 | 64))) {
    // the LDS-staged store epilogue writes 16-byte row chunks: rows 16-byte aligned and whole 8-column chunks inside the matrix,
    // otherwise the variant with the direct epilogue (which masks 4-column groups)
    const int nout = (EPI == VT_EPI_SWIGLU_BF16) ? (p.N >> 1) : p.N;
    if ((nout & 7) != 0 || (p.ldc & 7) != 0 || (((size_t)p.C) & 15) != 0) return launch_w4<EPI, MT, ABL | 64, AUX_A, AUX_B>(p, s);
  }
  auto kern = gemm_w4_kernel<EPI, MT, ABL, AUX_A, AUX_B>;
  w4_knobs_once();
  VT_LDS_ATTR_ONCE(kern, smem);
  hipLaunchKernelGGL(kern, dim3(cdiv(p.M, BM) * cdiv(p.N, 256) * (EPI == VT_EPI_F32 ? std::max(p.ksplit, 1) : 1)), dim3(256), smem, s, p);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

#include "vt_gemm8x.inc"   // gemm_w4x_kernel: the same tile with the MX-FP4 product of precise level 3 folded in

// ------------------------------------------------------------------------------------------------------------------
// gemm_w4r_kernel: the four-wave kernel on a SMALL tile, 160x128x64 (four waves of 80x64: 20 accumulator quads), for GEMMs whose
// big-tile grid would cover a fraction of the chip and whose 64x128 small-tile grid quantises badly -- the ViT's N = 1024
// projections: 4616 x 1024 = 232 tiles of 160x128 (one round, one per CU) against 584 tiles of 64x128 for 512 slots.
// A K step is only 40 MFMAs (0.35 us) here, far shorter than a trip to memory, so the LDS image is a RING OF FOUR K steps
// (4 x 36 KiB): the 9 LDS-DMA pieces a wave issues in the second k-half of K step t fill the buffer K step t has just finished
// with the data of K step t+4 -- three K steps (~1 us) of flight -- and ONE barrier per K step serves both hazards:
//     gap  0..8    read the 4 + 5 fragments of k-half 1 (buffer t % 4)
//     gap 17       lgkmcnt(0), vmcnt(18): this wave's reads of buffer t % 4 are done, its pieces of K step t+1 have landed
//     gap 18       barrier  -> buffer t % 4 is free for K step t+4 (WAR), buffer (t+1) % 4 is complete (RAW)
//     gap 20..23   read the B fragments of (t+1, k-half 0);  24, 26, .. 32: the A fragments
//     gap 25, 27, .. 33, 34 .. 37   the 9 pieces of K step t+4
//   (vmcnt: the pieces younger than K step t+1's are those of t+2 and t+3.) MFMA n: fragment set n / 20, accumulator row
//   (n % 20) / 4, column n % 4. Same fragments and accumulation order per output as every other tile kernel: identical bits.
// ------------------------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4r_kernel(GemmP8 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MT = 5, NI = 4, BM = 160, BN = 128;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUFB = A_BYTES + B_BYTES;   // 36 KiB per K step
  constexpr int NPA = 5, NPB = 4, NP = NPA + NPB;
  constexpr int H = MT * NI;                         // MFMAs per k-half
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  const int sid = xcd_remap((int)blockIdx.x, nwg);
  const int GROUP_M = W4_KNOB(1, 8);
  const int per_group = GROUP_M * tiles_n;
  const int first_m = (sid / per_group) * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (sid % per_group) % gsz;
  const int tn = (sid % per_group) / gsz;
  const int bm0 = tm * BM, bn0 = tn * BN;

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)bm0 * p.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)bn0 * p.ldw), 0, 0x7fffffff, 0x00020000);
  const int lrow = lane >> 3, lchk = lane & 7;
  int voff[9];   // pieces 0..3: B rows (4 wave + j) * 8 .., pieces 4..8: A rows (5 wave + j) * 8 ..
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = (i < NPB ? wave * NPB + i : wave * NPA + (i - NPB)) * 8 + lrow;
    const int coff = (lchk ^ ((row >> 1) & 7)) * 16;
    voff[i] = i < NPB ? (min(bn0 + row, p.N - 1) - bn0) * p.ldw * 2 + coff : (min(bm0 + row, p.M - 1) - bm0) * p.lda * 2 + coff;
  }
#define W4R_DMA(BUF, P, KB)                                                                                      \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                                      \
      (P) < NPB ? rsrc_b : rsrc_a,                                                                               \
      (__attribute__((address_space(3))) void*)(smem + (BUF) * BUFB +                                            \
                                                 ((P) < NPB ? A_BYTES + (wave * NPB + (P)) * 1024 : (wave * NPA + (P) - NPB) * 1024)), \
      16, voff[P], KB, 0, 0)

  const int f = (lane >> 1) & 7;
  const int fo[2] = {(lane & 15) * 128 + (((lane >> 4) ^ f) << 4), (lane & 15) * 128 + ((((lane >> 4) | 4) ^ f) << 4)};
  const int a_base = wr * (MT * 16 * 128);
  const int b_base = A_BYTES + wc * (NI * 16 * 128);

  f32x4 acc[MT][NI];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][MT], fb[2][NI];
#define W4R_MFMA(N)                                                                                              \
  asm volatile(VT_MFMA_16x16x32_ASM " %0, %1, %2, %0"                                                         \
               : "+a"(acc[((N) % H) / NI][(N) % NI]) : "v"(fb[((N) / H) & 1][(N) % NI]), "v"(fa[((N) / H) & 1][((N) % H) / NI]))
#define W4R_RDB(SET, BUF, KK, I) fb[SET][I] = *(const bf16x8*)(smem + (BUF) * BUFB + b_base + fo[KK] + (I) * 2048)
#define W4R_RDA(SET, BUF, KK, I) fa[SET][I] = *(const bf16x8*)(smem + (BUF) * BUFB + a_base + fo[KK] + (I) * 2048)

  const int nt = p.K >> 6;                         // K % 256 == 0, checked by the launcher
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int kb = min(b, nt - 1) * 128;
#pragma unroll
    for (int i = 0; i < NP; ++i) W4R_DMA(b, i, kb);
  }
  __builtin_amdgcn_s_waitcnt(w4_vmcnt_imm(3 * NP));   // K step 0 has landed
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NI; ++i) W4R_RDB(0, 0, 0, i);
#pragma unroll
  for (int i = 0; i < MT; ++i) W4R_RDA(0, 0, 0, i);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 4");                         // accumulator zeroes (v_accvgpr_write) -> first MFMA reading them as C

#define W4R_IX(x, m) ((((x) % (m)) + (m)) % (m))   /* keeps the indices of the untaken branches inside their arrays */
#define W4R_GAP(n, B, KB)                                                                                        \
  do {                                                                                                           \
    if ((n) < 4) W4R_RDB(1, B, 1, W4R_IX(n, 4));                                                                 \
    else if ((n) < 9) W4R_RDA(1, B, 1, W4R_IX((n) - 4, 5));                                                      \
    else if ((n) == 17) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_waitcnt(w4_vmcnt_imm(2 * NP)); } \
    else if ((n) == 18) __builtin_amdgcn_s_barrier();                                                            \
    else if ((n) >= 20 && (n) < 24) W4R_RDB(0, ((B) + 1) & 3, 0, W4R_IX((n) - 20, 4));                           \
    else if ((n) >= 24 && (n) <= 33) {                                                                           \
      if (!((n) & 1)) W4R_RDA(0, ((B) + 1) & 3, 0, W4R_IX(((n) - 24) >> 1, 5));                                  \
      else W4R_DMA(B, W4R_IX(((n) - 25) >> 1, 9), KB);                                                           \
    }                                                                                                            \
    else if ((n) >= 34 && (n) <= 37) W4R_DMA(B, 5 + W4R_IX((n) - 34, 4), KB);                                    \
  } while (0)
#define W4R_N(n, B, KB) W4R_MFMA(n); W4R_GAP(n, B, KB); __builtin_amdgcn_sched_barrier(0);
#define W4R_8(n, B, KB) W4R_N((n), B, KB) W4R_N((n) + 1, B, KB) W4R_N((n) + 2, B, KB) W4R_N((n) + 3, B, KB) \
                        W4R_N((n) + 4, B, KB) W4R_N((n) + 5, B, KB) W4R_N((n) + 6, B, KB) W4R_N((n) + 7, B, KB)
#define W4R_KSTEP(B, T)                                                                                          \
  do {                                                                                                           \
    const int _kb = min((T) + 4, nt - 1) * 128;    /* past the end of K: re-fetch the last K step, harmless */    \
    W4R_8(0, B, _kb) W4R_8(8, B, _kb) W4R_8(16, B, _kb) W4R_8(24, B, _kb) W4R_8(32, B, _kb)                       \
  } while (0)
  for (int t = 0; t < nt; t += 4) {
    W4R_KSTEP(0, t);
    W4R_KSTEP(1, t + 1);
    W4R_KSTEP(2, t + 2);
    W4R_KSTEP(3, t + 3);
    VT_MFMA_TAIL_WAIT(t + 4 >= nt);
  }
  VT_VMCNT(0);                                     // the tail's redundant pieces
  asm volatile("s_nop 15\n\ts_nop 7");            // last MFMA result -> the epilogue's v_accvgpr_read
#undef W4R_KSTEP
#undef W4R_8
#undef W4R_N
#undef W4R_GAP
#undef W4R_IX
#undef W4R_RDA
#undef W4R_RDB
#undef W4R_MFMA
#undef W4R_DMA
  w4_epilogue<EPI, MT, NI>(p, acc, bm0, bn0, wr, wc, lane);
}

template <int EPI>
int launch_w4r(const GemmP8& p, hipStream_t s) {
  constexpr int smem = 4 * (160 + 128) * 128;      // 144 KiB
  w4_knobs_once();
  auto kern = gemm_w4r_kernel<EPI>;
  VT_LDS_ATTR_ONCE(kern, smem);
  hipLaunchKernelGGL(kern, dim3(cdiv(p.M, 160) * cdiv(p.N, 128)), dim3(256), smem, s, p);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// gemm_rp_kernel ("register-pipelined"): 256x256x64 tile, 8 waves (2 x 4, two per SIMD), 128x64 per wave as 4 x 2
// fragments of v_mfma_f32_32x32x16_bf16. No load/compute phase split: inside every wave the fragment reads of k-step
// j+1 (6 ds_read_b128) are in flight while the 8 MFMAs of k-step j issue (8 independent accumulators, no dependent
// back-to-back MFMAs), fragments double-buffered in registers. Two LDS stages, LDS-DMA of K tile t+1 issued right after
// the barrier, ONE barrier per K tile placed before the last k-step so the first fragments of tile t+1 are read under the
// last MFMAs of tile t.
// ------------------------------------------------------------------------------------------------------------------
// VAR bit0: fragment reads 2 per MFMA (front-loaded) instead of 1 per MFMA; bit1: DMA of tile t+2 right after the barrier of
// tile t (a full K tile ahead) instead of tile t+1 at the top of iteration t. (A/B knobs: tools/gemm_bench.py)
template <int EPI, int VAR = 0>
__global__ __launch_bounds__(512, 2) void gemm_rp_kernel(GemmP8 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE_BYTES = 256 * 64 * 2;
  constexpr int STAGE = 2 * TILE_BYTES;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
  const int nwg = tiles_m * tiles_n;
  const int sid = xcd_remap(blockIdx.x, nwg);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int first_m = (sid / per_group) * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (sid % per_group) % gsz;
  const int tn = (sid % per_group) / gsz;
  const int bm0 = tm * 256, bn0 = tn * 256;

  // DMA: 32 pieces (8 rows each) per operand tile; wave w stages pieces 4w..4w+3 of A and of B
  const int lrow = lane >> 3, lchk = lane & 7;
  unsigned a_src[4], b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + lrow;
    const int coff = (lchk ^ ((row >> 1) & 7)) * 8;
    a_src[i] = (unsigned)(((size_t)min(bm0 + row, p.M - 1) * p.lda + coff) * 2);
    b_src[i] = (unsigned)(((size_t)min(bn0 + row, p.N - 1) * p.ldw + coff) * 2);
  }
  const char* a_ptr = (const char*)p.A;
  const char* b_ptr = (const char*)p.W;
  const int dma_off = wave * 4096;
#define RP_STAGE(BUF)                                                        \
  do {                                                                       \
    char* _d = smem + (BUF) * STAGE + dma_off;                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                          \
      glds16(a_ptr + a_src[i], _d + i * 1024);                               \
      a_src[i] += 128;                                                       \
    }                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                          \
      glds16(b_ptr + b_src[i], _d + TILE_BYTES + i * 1024);                  \
      b_src[i] += 128;                                                       \
    }                                                                        \
  } while (0)

  const int f = ((lane & 31) >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fo[ks] = (lane & 31) * 128 + ((((2 * ks) | (lane >> 5)) ^ f) << 4);
  const int a_base = wr * 128 * 128;
  const int b_base = TILE_BYTES + wc * 64 * 128;

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];

#define RP_READ(FA, FB, BUF, KS)                                                          \
  do {                                                                                    \
    const char* _s = smem + (BUF) * STAGE;                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) FA[i] = *(const bf16x8*)(_s + a_base + i * 4096 + fo[KS]); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) FB[i] = *(const bf16x8*)(_s + b_base + i * 4096 + fo[KS]); \
  } while (0)
#define RP_MFMA(FA, FB)                                                                   \
  do {                                                                                    \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                      \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                    \
        acc[mi][ni] = VT_MFMA_32x32x16(FB[ni], FA[mi], acc[mi][ni]); \
  } while (0)
  // one k-step: reads of the next fragments interleaved 1:1 behind the first MFMAs
#define RP_STEP(FA_CUR, FB_CUR, FA_NXT, FB_NXT, BUF, KS_NXT)                              \
  do {                                                                                    \
    RP_READ(FA_NXT, FB_NXT, BUF, KS_NXT);                                                 \
    RP_MFMA(FA_CUR, FB_CUR);                                                              \
    if (VAR & 1) {                                                                        \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                \
      }                                                                                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);                                  \
    } else {                                                                              \
      _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                \
      }                                                                                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                  \
    }                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)

  const int nt = p.K >> 6;
  constexpr bool EARLY = (VAR & 2) != 0;
  RP_STAGE(0);
  if (EARLY && nt > 1) RP_STAGE(1);
  if (EARLY && nt > 1) VT_VMCNT(8); else VT_VMCNT(0);   // K tile 0 landed (EARLY: tile 1's 8 DMAs may still be in flight)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  RP_READ(fa0, fb0, 0, 0);

  // K tile t: k-steps 1..3 read buffer t&1; then "tile t+1 landed" wait + barrier (buffer t&1 is dead from here on) and the
  // last k-step already reads tile t+1's first fragments. DMA: EARLY -> tile t+2 into the dead buffer right after the barrier
  // (a full K tile of lead); otherwise tile t+1 at the top of iteration t (three k-steps of lead).
  // STAGE_ON / NEXT_ON are compile-time: the steady-state body has no branches.
#define RP_TILE(STAGE_ON, NEXT_ON)                                                         \
  do {                                                                                    \
    const int cur = t & 1;                                                                \
    if (!EARLY && NEXT_ON) RP_STAGE(cur ^ 1);                                             \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    RP_STEP(fa0, fb0, fa1, fb1, cur, 1);                                                  \
    RP_STEP(fa1, fb1, fa0, fb0, cur, 2);                                                  \
    RP_STEP(fa0, fb0, fa1, fb1, cur, 3);                                                  \
    if (NEXT_ON) {                                                                        \
      VT_VMCNT(0);                                                                        \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                  \
      __builtin_amdgcn_s_barrier();                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                  \
      if (EARLY && STAGE_ON) RP_STAGE(cur);                                               \
      __builtin_amdgcn_sched_barrier(0);                                                  \
      RP_STEP(fa1, fb1, fa0, fb0, cur ^ 1, 0);                                            \
    } else {                                                                              \
      RP_MFMA(fa1, fb1);                                                                  \
    }                                                                                     \
  } while (0)
  int t = 0;
  for (; t + 2 < nt; ++t) RP_TILE(true, true);
  if (t + 1 < nt) {
    RP_TILE(false, true);
    ++t;
  }
  RP_TILE(false, false);
#undef RP_TILE

  // ---- epilogue: lane (m = ..+(lane&31)) holds n = nfrag + 8*g + 4*(lane>>5) + {0..3}, g = reg>>2 -------------------------
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = bm0 + wr * 128 + mi * 32 + (lane & 31);
    if (m >= p.M) continue;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int nfrag = bn0 + wc * 64 + ni * 32;
      const f32x16 a = acc[mi][ni];
      if constexpr (EPI == VT_EPI_SWIGLU_BF16) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int nl = 8 * g + 4 * (lane >> 5);
          if (nfrag + nl >= p.N) continue;
          u32x2 o;
          o.x = pack_op2(silu8(a[4 * g + 0]) * a[4 * g + 8], silu8(a[4 * g + 1]) * a[4 * g + 9]);
          o.y = pack_op2(silu8(a[4 * g + 2]) * a[4 * g + 10], silu8(a[4 * g + 3]) * a[4 * g + 11]);
          *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + (nfrag >> 1) + nl) = o;
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nfrag + 8 * g + 4 * (lane >> 5);
          if (n >= p.N) continue;
          f32x4 v = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
          if (p.bias) v += *(const f32x4*)(p.bias + n);
          if constexpr (EPI == VT_EPI_BF16_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf8(v[r]);
          } else if constexpr (EPI == VT_EPI_BF16_QGELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = quick_gelu8(v[r]);
          } else if constexpr (EPI == VT_EPI_BF16_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          if constexpr (EPI == VT_EPI_F32_RESID) {
            float* c = (float*)p.C + (size_t)m * p.ldc + n;
            *(f32x4*)c = *(const f32x4*)c + v;
          } else if constexpr (EPI == VT_EPI_F32) {
            *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = v;
          } else {
            u32x2 o;
            o.x = pack_op2(v[0], v[1]);
            o.y = pack_op2(v[2], v[3]);
            *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
          }
        }
      }
    }
  }
}

template <int EPI, int VAR = 0>
int launch_rp(const GemmP8& p, hipStream_t s) {
  constexpr int smem = 2 * 2 * 256 * 64 * 2;  // 128 KiB
  auto kern = gemm_rp_kernel<EPI, VAR>;
  VT_LDS_ATTR_ONCE(kern, smem);
  const int nwg = cdiv(p.M, 256) * cdiv(p.N, 256) * (p.ksplit > 1 ? p.ksplit : 1);
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), smem, s, p);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

template <int EPI, int ABL = 0, bool P4 = false>
int launch_p8(const GemmP8& p, hipStream_t s) {
  constexpr int smem = 2 * BUF_BYTES;  // 128 KiB
  auto kern = gemm_p8_kernel<EPI, ABL, P4>;
  VT_LDS_ATTR_ONCE(kern, smem);
  const int nwg = cdiv(p.M, 256) * cdiv(p.N, 256) * (p.ksplit > 1 ? p.ksplit : 1);
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), smem, s, p);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

}  // namespace

bool vt_gemm_p8_supported(int /*M*/, int N, int K) { return (K % 128) == 0 && K >= 256 && N % 32 == 0; }

int vt_gemm_rp_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, const float* bias, int M,
                      int N, int K, int epi, hipStream_t s) {
  VT_REQUIRE((K % 64) == 0 && N % 32 == 0, "vt_gemm(rp): needs K %% 64 == 0 and N %% 32 == 0 (K=%d N=%d)", K, N);
  VT_REQUIRE((size_t)M * lda * 2 < (1ull << 32) && (size_t)N * ldw * 2 < (1ull << 32), "vt_gemm(rp): operands must be < 4 GiB");
  GemmP8 p{A, W, C, bias, M, N, K, lda, ldw, ldc, 1, 0, VtGemmNormFuse{}};
#ifdef VT_ABLATIONS
  if (epi >= 0x100) {  // A/B variants of the main loop (bf16 epilogue only); test library only
    switch (epi >> 8) {
      case 1: return launch_rp<VT_EPI_BF16, 1>(p, s);
      case 2: return launch_rp<VT_EPI_BF16, 2>(p, s);
      default: return launch_rp<VT_EPI_BF16, 3>(p, s);
    }
  }
#endif
  switch (epi) {
    case VT_EPI_BF16: return launch_rp<VT_EPI_BF16>(p, s);
    case VT_EPI_BF16_GELU: return launch_rp<VT_EPI_BF16_GELU>(p, s);
    case VT_EPI_BF16_QGELU: return launch_rp<VT_EPI_BF16_QGELU>(p, s);
    case VT_EPI_BF16_RELU: return launch_rp<VT_EPI_BF16_RELU>(p, s);
    case VT_EPI_F32_RESID: return launch_rp<VT_EPI_F32_RESID>(p, s);
    case VT_EPI_F32: return launch_rp<VT_EPI_F32>(p, s);
    case VT_EPI_SWIGLU_BF16: return launch_rp<VT_EPI_SWIGLU_BF16>(p, s);
    default: vt_set_error("vt_gemm(rp): unknown epilogue %d", epi); return VT_ERR_ARG;
  }
}

// Residual GEMM as a two-pass split-K: pass 1 = the 4-phase ping-pong kernel over `ksplit` K ranges, fp32 partial products to
// `partials` ([ksplit][M][N]); pass 2 adds them (split order: deterministic) and the bias-carrying split 0 onto the residual C.
namespace {
__global__ __launch_bounds__(256) void splitk_reduce_resid_kernel(const float* __restrict__ part, size_t slab, int ksplit,
                                                                  float* __restrict__ C, int ldc, int M, int N, VtGemmNormFuse nf) {
  const int n4 = N >> 2;
  // folded RMSNorm producer (N % 256 == 0 there): 8 consecutive lanes own one 32-column group of one row
  const long total = (long)M * n4;
  const long total_r = nf.out_partials ? ((total + 63) & ~63L) : total;   // whole waves enter the shuffles
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_r; i += (long)gridDim.x * blockDim.x) {
    const bool live = i < total;
    const int m = live ? (int)(i / n4) : 0, c = live ? (int)(i % n4) : 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      acc = *(const f32x4*)(C + (size_t)m * ldc + c * 4);
      for (int s = 0; s < ksplit; ++s) acc += *(const f32x4*)(part + (size_t)s * slab + (size_t)m * N + c * 4);
      *(f32x4*)(C + (size_t)m * ldc + c * 4) = acc;
    }
    if (nf.out_partials) {
      float ss = (acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3]);
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      ss += __shfl_xor(ss, 4, 64);
      if (live) {
        const f32x4 w4 = *(const f32x4*)(nf.out_w + c * 4);
        u32x2 o;
        o.x = pack_op2(acc[0] * w4[0], acc[1] * w4[1]);
        o.y = pack_op2(acc[2] * w4[2], acc[3] * w4[3]);
        *(u32x2*)(nf.out_xw + (size_t)m * nf.ld_xw + c * 4) = o;
        if ((c & 7) == 0) nf.out_partials[(size_t)(c >> 3) * nf.out_ldp + m] = ss;
      }
    }
  }
}
}  // namespace

int vt_gemm_p4_splitk_resid_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, float* C, int ldc, const float* bias, int M,
                                   int N, int K, int ksplit, float* partials, hipStream_t s, const VtGemmNormFuse* nf) {
  VT_REQUIRE(vt_gemm_p8_supported(M, N, K) && (N % 4) == 0, "vt_gemm(split-K): unsupported shape (N=%d K=%d)", N, K);
  VT_REQUIRE(ksplit >= 2 && (K >> 7) / ksplit >= 2 && partials, "vt_gemm(split-K): ksplit=%d leaves < 256 of K per split, or no workspace", ksplit);
  GemmP8 p{A, W, partials, bias, M, N, K, lda, ldw, N, ksplit, (size_t)M * N, VtGemmNormFuse{}};   // the fold happens in the reduce pass
  // pass 1 on the four-wave kernel (round 3; was the ping-pong kernel), on the tile height that pads fewer rows
  if ((long)cdiv(M, 224) * 224 < (long)cdiv(M, 256) * 256) VT_TRY((launch_w4<VT_EPI_F32, 7>(p, s)));
  else VT_TRY((launch_w4<VT_EPI_F32, 8>(p, s)));
  const long total = (long)M * (N >> 2);
  const int blocks = (int)std::min<long>((total + 255) / 256, 2048);
  VtGemmNormFuse rnf;
  if (nf && nf->out_partials) {
    VT_REQUIRE((N % 32) == 0 && nf->out_np >= N / 32 && nf->out_ldp >= M && nf->out_w && nf->out_xw, "vt_gemm(split-K): norm fold needs N %% 32 == 0 and its buffers");
    rnf = *nf;
  }
  hipLaunchKernelGGL(splitk_reduce_resid_kernel, dim3(blocks), dim3(256), 0, s, partials, p.slab, ksplit, C, ldc, M, N, rnf);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

// ---- fused QKV epilogue: host side ------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void row_slot_kernel(const VtAttnSeq* __restrict__ seqs, const int* __restrict__ tile_table,
                                                       int* __restrict__ row_slot) {
  const VtAttnSeq sq = seqs[blockIdx.y];
  const int past = sq.kv_len - sq.q_len;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < sq.q_len; i += gridDim.x * blockDim.x) {
    const int pos = past + i;
    const int page = tile_table[sq.table_off + (pos >> 6)];
    const int tail = (i == sq.q_len - 1) ? 63 - (pos & 63) : 0;
    row_slot[sq.q_row0 + i] = (tail << 26) | (page * 64 + (pos & 63));
  }
}
}  // namespace

int vt_row_slot_launch(const VtAttnSeq* seqs, int nseq, int max_q_len, const int* tile_table, int* row_slot, hipStream_t s) {
  VT_REQUIRE(seqs && tile_table && row_slot && nseq > 0 && max_q_len > 0, "vt_row_slot: bad arguments");
  hipLaunchKernelGGL(row_slot_kernel, dim3(std::min(cdiv(max_q_len, 256), 64), nseq), dim3(256), 0, s, seqs, tile_table, row_slot);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

bool vt_gemm_qkv_fused_supported(int rows, int H, int head_dim) {
  return head_dim == 128 && (H % 256) == 0 && rows > 64 && vt_gemm_p8_supported(rows, 3 * H, H) &&
         vt_gemm_pick_cfg(rows, 3 * H, H) == VT_GEMM_CFG_256x256_P4;
}

int vt_gemm_qkv_fused_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, bf16_t* qkv, int ldqkv, int M, int H,
                             const VtQkvFuse& qf, hipStream_t s) {
  VT_REQUIRE(A && W && qkv && qf.k_pages && qf.vt_pages && qf.row_slot && qf.positions && qf.rope_cos && qf.rope_sin,
             "vt_gemm(qkv fused): null pointer");
  VT_REQUIRE(qf.heads * 128 == H && vt_gemm_qkv_fused_supported(M, H, 128), "vt_gemm(qkv fused): unsupported shape (M=%d H=%d)", M, H);
  VT_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0 && (ldqkv % 8) == 0, "vt_gemm(qkv fused): misaligned leading dimensions");
  GemmP8 p{A, W, qkv, nullptr, M, 3 * H, H, lda, ldw, ldqkv, 1, 0, VtGemmNormFuse{}, qf};
  VtProfScope prof(VT_PROF_GEMM_TILE, 2.0 * (double)M * 3.0 * (double)H * (double)H, s);
  return launch_p8<VT_EPI_QKV_PAGES, 0, true>(p, s);
}

int vt_gemm_p8_launch(const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, const float* bias, int M,
                      int N, int K, int epi, hipStream_t s, const VtGemmNormFuse* nf) {
  VT_REQUIRE(vt_gemm_p8_supported(M, N, K), "vt_gemm(p8): needs K %% 128 == 0, K >= 256, N %% 32 == 0 (K=%d N=%d)", K, N);
  GemmP8 p{A, W, C, bias, M, N, K, lda, ldw, ldc, 1, 0, nf ? *nf : VtGemmNormFuse{}};
  if (p.nf.out_partials)
    VT_REQUIRE((epi & 0xff) == VT_EPI_F32_RESID && p.nf.out_w && p.nf.out_xw && p.nf.out_np >= N / 32 && p.nf.out_ldp >= M,
               "vt_gemm(p8): norm-fold producer needs the residual epilogue and its buffers");
#ifdef VT_ABLATIONS
  if (epi >= 0x100 && !(epi & 0x1d000)) {  // timing ablations (tools/gemm_ablate.py); test library only
    switch (epi >> 8) {
      case 1: return launch_p8<VT_EPI_BF16, 1>(p, s);
      case 2: return launch_p8<VT_EPI_BF16, 2>(p, s);
      case 3: return launch_p8<VT_EPI_BF16, 3>(p, s);
      case 4: return launch_p8<VT_EPI_BF16, 4>(p, s);
      case 5: return launch_p8<VT_EPI_BF16, 5>(p, s);
      case 6: return launch_p8<VT_EPI_BF16, 6>(p, s);
      default: return launch_p8<VT_EPI_BF16, 7>(p, s);
    }
  }
  if ((epi & 0x1000) && (epi & 0x700)) {   // timing ablations of the 4-phase loop
    switch ((epi >> 8) & 7) {
      case 1: return launch_p8<VT_EPI_BF16, 1, true>(p, s);
      case 2: return launch_p8<VT_EPI_BF16, 2, true>(p, s);
      case 3: return launch_p8<VT_EPI_BF16, 3, true>(p, s);
      case 4: return launch_p8<VT_EPI_BF16, 4, true>(p, s);
      case 5: return launch_p8<VT_EPI_BF16, 5, true>(p, s);
      case 6: return launch_p8<VT_EPI_BF16, 6, true>(p, s);
      default: return launch_p8<VT_EPI_BF16, 7, true>(p, s);
    }
  }
#endif
  if (epi & 0x10000) {   // 4-wave kernel, 160x128 tile, four-deep ring
    VT_REQUIRE(!nf && (K % 256) == 0, "vt_gemm(w4r): needs K %% 256 == 0 and no norm fold (K=%d)", K);
    switch (epi & 0xff) {
      case VT_EPI_BF16: return launch_w4r<VT_EPI_BF16>(p, s);
      case VT_EPI_BF16_GELU: return launch_w4r<VT_EPI_BF16_GELU>(p, s);
      case VT_EPI_BF16_QGELU: return launch_w4r<VT_EPI_BF16_QGELU>(p, s);
      case VT_EPI_BF16_RELU: return launch_w4r<VT_EPI_BF16_RELU>(p, s);
      case VT_EPI_F32_RESID: return launch_w4r<VT_EPI_F32_RESID>(p, s);
      case VT_EPI_F32: return launch_w4r<VT_EPI_F32>(p, s);
      case VT_EPI_SWIGLU_BF16: return launch_w4r<VT_EPI_SWIGLU_BF16>(p, s);
      default: vt_set_error("vt_gemm(w4r): unknown epilogue %d", epi & 0xff); return VT_ERR_ARG;
    }
  }
#ifdef VT_ABLATIONS
  static const int w4_direct = getenv("VT_W4_EPI_DIRECT") ? atoi(getenv("VT_W4_EPI_DIRECT")) : 0;   // A/B: the direct store epilogue
  if (w4_direct && (epi & 0x4000)) {
    const bool t320 = (epi & 0x8000) != 0;
    switch (epi & 0xff) {
      case VT_EPI_BF16: return t320 ? launch_w4<VT_EPI_BF16, 10, 64>(p, s) : launch_w4<VT_EPI_BF16, 8, 64>(p, s);
      case VT_EPI_BF16_GELU: return t320 ? launch_w4<VT_EPI_BF16_GELU, 10, 64>(p, s) : launch_w4<VT_EPI_BF16_GELU, 8, 64>(p, s);
      case VT_EPI_SWIGLU_BF16: return t320 ? launch_w4<VT_EPI_SWIGLU_BF16, 10, 64>(p, s) : launch_w4<VT_EPI_SWIGLU_BF16, 8, 64>(p, s);
      default: break;
    }
  }
#endif
  if ((epi & 0x4000) && (epi & 0x20000)) {   // 4-wave kernel, 224-row tile (112x128 per wave)
    VT_REQUIRE(!nf, "vt_gemm(w4): no norm fold in the 4-wave kernel");
    switch (epi & 0xff) {
      case VT_EPI_BF16: return launch_w4<VT_EPI_BF16, 7>(p, s);
      case VT_EPI_BF16_GELU: return launch_w4<VT_EPI_BF16_GELU, 7>(p, s);
      case VT_EPI_BF16_QGELU: return launch_w4<VT_EPI_BF16_QGELU, 7>(p, s);
      case VT_EPI_BF16_RELU: return launch_w4<VT_EPI_BF16_RELU, 7>(p, s);
      case VT_EPI_F32_RESID: return launch_w4<VT_EPI_F32_RESID, 7>(p, s);
      case VT_EPI_F32: return launch_w4<VT_EPI_F32, 7>(p, s);
      case VT_EPI_SWIGLU_BF16: return launch_w4<VT_EPI_SWIGLU_BF16, 7>(p, s);
      default: vt_set_error("vt_gemm(w4, 224-row tile): epilogue %d not instantiated", epi & 0xff); return VT_ERR_ARG;
    }
  }
  if ((epi & 0x4000) && (epi & 0x8000)) {   // 4-wave kernel, 320-row tile (160x128 per wave)
    VT_REQUIRE(!nf, "vt_gemm(w4): no norm fold in the 4-wave kernel");
    switch (epi & 0xff) {
      case VT_EPI_BF16: return launch_w4<VT_EPI_BF16, 10>(p, s);
      case VT_EPI_BF16_GELU: return launch_w4<VT_EPI_BF16_GELU, 10>(p, s);
      case VT_EPI_BF16_QGELU: return launch_w4<VT_EPI_BF16_QGELU, 10>(p, s);
      case VT_EPI_BF16_RELU: return launch_w4<VT_EPI_BF16_RELU, 10>(p, s);
      case VT_EPI_F32_RESID: return launch_w4<VT_EPI_F32_RESID, 10>(p, s);
      case VT_EPI_F32: return launch_w4<VT_EPI_F32, 10>(p, s);
      case VT_EPI_SWIGLU_BF16: return launch_w4<VT_EPI_SWIGLU_BF16, 10>(p, s);
      default: vt_set_error("vt_gemm(w4, 320-row tile): epilogue %d not instantiated", epi & 0xff); return VT_ERR_ARG;
    }
  }
  if (epi & 0x4000) {   // 4-wave kernel (128x128 per wave)
    VT_REQUIRE(!nf, "vt_gemm(w4): no norm fold in the 4-wave kernel");
    switch (epi & 0xff) {
      case VT_EPI_BF16: {
#ifdef VT_ABLATIONS
        static const int abl = getenv("VT_W4_ABL") ? atoi(getenv("VT_W4_ABL")) : 0;   // timing ablations, see the kernel
        switch (abl) {
          case 1: return launch_w4<VT_EPI_BF16, 8, 1>(p, s);
          case 2: return launch_w4<VT_EPI_BF16, 8, 2>(p, s);
          case 4: return launch_w4<VT_EPI_BF16, 8, 4>(p, s);
          case 5: return launch_w4<VT_EPI_BF16, 8, 5>(p, s);
          case 8: return launch_w4<VT_EPI_BF16, 8, 8>(p, s);
          case 15: return launch_w4<VT_EPI_BF16, 8, 15>(p, s);
          case 16: return launch_w4<VT_EPI_BF16, 8, 16>(p, s);
          case 32: return launch_w4<VT_EPI_BF16, 8, 32>(p, s);     // no epilogue stores
          case 64: return launch_w4<VT_EPI_BF16, 8, 64>(p, s);     // the direct (8 bytes per lane) store epilogue
          default: break;
        }
#endif
        return launch_w4<VT_EPI_BF16>(p, s);
      }
      case VT_EPI_BF16_GELU: return launch_w4<VT_EPI_BF16_GELU>(p, s);
      case VT_EPI_BF16_QGELU: return launch_w4<VT_EPI_BF16_QGELU>(p, s);
      case VT_EPI_BF16_RELU: return launch_w4<VT_EPI_BF16_RELU>(p, s);
      case VT_EPI_F32_RESID: return launch_w4<VT_EPI_F32_RESID>(p, s);
      case VT_EPI_F32: return launch_w4<VT_EPI_F32>(p, s);
      case VT_EPI_SWIGLU_BF16: return launch_w4<VT_EPI_SWIGLU_BF16>(p, s);
      default: vt_set_error("vt_gemm(w4): epilogue %d not instantiated", epi & 0xff); return VT_ERR_ARG;
    }
  }
  if (epi & 0x1000) {   // 4-phase variant
    switch (epi & 0xff) {
      case VT_EPI_BF16: return launch_p8<VT_EPI_BF16, 0, true>(p, s);
      case VT_EPI_BF16_GELU: return launch_p8<VT_EPI_BF16_GELU, 0, true>(p, s);
      case VT_EPI_BF16_QGELU: return launch_p8<VT_EPI_BF16_QGELU, 0, true>(p, s);
      case VT_EPI_BF16_RELU: return launch_p8<VT_EPI_BF16_RELU, 0, true>(p, s);
      case VT_EPI_F32_RESID: return launch_p8<VT_EPI_F32_RESID, 0, true>(p, s);
      case VT_EPI_F32: return launch_p8<VT_EPI_F32, 0, true>(p, s);
      case VT_EPI_SWIGLU_BF16: return launch_p8<VT_EPI_SWIGLU_BF16, 0, true>(p, s);
      default: vt_set_error("vt_gemm(p4): unknown epilogue %d", epi & 0xff); return VT_ERR_ARG;
    }
  }
  switch (epi) {
    case VT_EPI_BF16: return launch_p8<VT_EPI_BF16>(p, s);
    case VT_EPI_BF16_GELU: return launch_p8<VT_EPI_BF16_GELU>(p, s);
    case VT_EPI_BF16_QGELU: return launch_p8<VT_EPI_BF16_QGELU>(p, s);
    case VT_EPI_BF16_RELU: return launch_p8<VT_EPI_BF16_RELU>(p, s);
    case VT_EPI_F32_RESID: return launch_p8<VT_EPI_F32_RESID>(p, s);
    case VT_EPI_F32: return launch_p8<VT_EPI_F32>(p, s);
    case VT_EPI_SWIGLU_BF16: return launch_p8<VT_EPI_SWIGLU_BF16>(p, s);
    default: vt_set_error("vt_gemm(p8): unknown epilogue %d", epi); return VT_ERR_ARG;
  }
}

// precise level 3: C = epi(A.W^T + (A4 2^ea).(W4 2^ew)^T) on 256x256 tiles (vt_gemm8x.inc). M, N arbitrary (edges clamped / masked as in the
// base kernel), K % 128 == 0 and K >= 256; aexp must cover ceil(M / 256) * 256 rows (vt_mx4_aexp_bytes). ksplit > 1: EPI F32 only, split s
// of every tile writes slab s (C + s * slab elements). epi = VT_EPI_SWIGLU_MX (internal): out4 / oexp receive the output's own 4-bit image.
int vt_gemm_mx_launch(const bf16_t* A, int lda, const uint8_t* A4, const uint8_t* aexp, const bf16_t* W, int ldw, const uint8_t* W4,
                      const uint8_t* wexp, void* C, int ldc, const float* bias, int M, int N, int K, int epi, int ksplit, size_t slab,
                      uint8_t* out4, uint8_t* oexp, hipStream_t s) {
  VT_REQUIRE(A && A4 && aexp && W && W4 && wexp && C, "vt_gemm_mx: null pointer");
  VT_REQUIRE(M > 0 && N > 0 && (K % 128) == 0 && K >= 256 && (N % 4) == 0, "vt_gemm_mx: needs K %% 128 == 0, K >= 256, N %% 4 == 0 (K=%d N=%d)", K, N);
  VT_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0 && (((size_t)A4 | (size_t)W4) & 15) == 0, "vt_gemm_mx: misaligned operands");
  VT_REQUIRE(ksplit <= 1 || (epi == VT_EPI_F32 && (K / 128) >= 2 * ksplit), "vt_gemm_mx: split-K needs the fp32 epilogue and >= 2 K pairs per split");
  GemmP8 p{A, W, C, bias, M, N, K, lda, ldw, ldc, std::max(ksplit, 1), slab, VtGemmNormFuse{}};
  const GemmMx x{A4, aexp, W4, wexp, out4, oexp};
  VtProfScope prof(VT_PROF_GEMM_TILE, 2.0 * (double)M * (double)N * (double)K, s);
  switch (epi) {
    case VT_EPI_BF16: return launch_w4x<VT_EPI_BF16>(p, x, s);
    case VT_EPI_BF16_GELU: return launch_w4x<VT_EPI_BF16_GELU>(p, x, s);
    case VT_EPI_BF16_QGELU: return launch_w4x<VT_EPI_BF16_QGELU>(p, x, s);
    case VT_EPI_F32_RESID: return launch_w4x<VT_EPI_F32_RESID>(p, x, s);
    case VT_EPI_F32: return launch_w4x<VT_EPI_F32>(p, x, s);
    case VT_EPI_SWIGLU_BF16: return launch_w4x<VT_EPI_SWIGLU_BF16>(p, x, s);
    case VT_EPI_SWIGLU_MX:
      VT_REQUIRE(out4 && oexp && (N % 128) == 0 && (ldc % 8) == 0 && (((size_t)C | (size_t)out4) & 15) == 0 && !bias,
                 "vt_gemm_mx: the SwiGLU epilogue with the level 3 operand out needs N %% 128 == 0, aligned outputs and no bias");
      return launch_w4x<VT_EPI_SWIGLU_MX>(p, x, s);
    case VT_EPI_GELU_MX:
    case VT_EPI_QGELU_MX:
      VT_REQUIRE(out4 && oexp && (N % 64) == 0 && (ldc % 8) == 0 && (((size_t)C | (size_t)out4) & 15) == 0,
                 "vt_gemm_mx: the GELU epilogues with the level 3 operand out need N %% 64 == 0 and aligned outputs");
      return epi == VT_EPI_GELU_MX ? launch_w4x<VT_EPI_GELU_MX>(p, x, s) : launch_w4x<VT_EPI_QGELU_MX>(p, x, s);
    default: vt_set_error("vt_gemm_mx: epilogue %d not instantiated", epi); return VT_ERR_ARG;
  }
}

// the residual GEMMs of level 3 (x += A.W^T + A4.W4^T: o_proj, down_proj). The kernel has ONE tile height (256 rows; the 4-bit image takes the
// LDS the 320-row tile would need), so a grid that spills a fraction of a round over whole rounds of the 256 CUs -- 5120 x 4096: 320 tiles --
// is cut at the last whole round: the leading row blocks run as they are, the trailing ones as `ksplit` K ranges per tile on the otherwise
// idle CUs (fp32 partial slabs + the ordered reduce of the two-pass split-K), 1.25 rounds + a reduce instead of 2.
int vt_gemm_mx_resid_launch(const bf16_t* A, int lda, const uint8_t* A4, const uint8_t* aexp, const bf16_t* W, int ldw, const uint8_t* W4,
                            const uint8_t* wexp, float* C, int ldc, const float* bias, int M, int N, int K, float* partials, size_t partial_bytes,
                            hipStream_t s) {
  const int tiles_m = cdiv(M, 256), tiles_n = cdiv(N, 256), total = tiles_m * tiles_n, CUS = 256;
  int head_rows = (total / CUS) * CUS / tiles_n;              // row blocks inside whole rounds
  // a grid of at most half a round (the towers' fc2: 4616 x 1024 = 76 tiles) has no head: the whole problem runs as K ranges
  if (total * 2 <= CUS) head_rows = 0;
  const int tail_tiles = (tiles_m - head_rows) * tiles_n;
  int ksplit = tail_tiles > 0 ? std::min(8, CUS / tail_tiles) : 1;
  while (ksplit > 1 && (K / 128) < 2 * ksplit) --ksplit;
  const int m0 = head_rows * 256, mt = M - m0;
  if (tail_tiles == 0 || ksplit < 2 || !partials || (size_t)ksplit * mt * N * 4 > partial_bytes || (N % 4) != 0 || (head_rows == 0 && total * 2 > CUS))
    return vt_gemm_mx_launch(A, lda, A4, aexp, W, ldw, W4, wexp, C, ldc, bias, M, N, K, VT_EPI_F32_RESID, 1, 0, nullptr, nullptr, s);
  if (head_rows > 0)
    VT_TRY(vt_gemm_mx_launch(A, lda, A4, aexp, W, ldw, W4, wexp, C, ldc, bias, m0, N, K, VT_EPI_F32_RESID, 1, 0, nullptr, nullptr, s));
  const size_t slab = (size_t)mt * N;
  VT_TRY(vt_gemm_mx_launch(A + (size_t)m0 * lda, lda, A4 + (size_t)m0 * (K >> 1), aexp + (size_t)(m0 >> 6) * (K >> 5) * 64, W, ldw, W4, wexp,
                           partials, N, bias, mt, N, K, VT_EPI_F32, ksplit, slab, nullptr, nullptr, s));    // (split 0 adds the bias)
  const long tot4 = (long)mt * (N >> 2);
  hipLaunchKernelGGL(splitk_reduce_resid_kernel, dim3((int)std::min<long>((tot4 + 255) / 256, 2048)), dim3(256), 0, s, partials, slab, ksplit,
                     C + (size_t)m0 * ldc, ldc, mt, N, VtGemmNormFuse{});
  VT_LAUNCH_CHECK();
  return VT_OK;
}
