// mx_probe -- what v_mfma_scale_f32_16x16x128_f8f6f4 does with FP4 (e2m1) operands on gfx950, and what it costs next to the
// 16-bit MFMA the GEMM main loops are built from. Measurement helper behind the mixed-precision GEMM (vt_gemm8x.hip): the
// lo halves of operand pairs run on the MX pipe, so the operand layout, the nibble order, the e8m0 scale semantics and the issue
// cost of the instruction are pinned here on the hardware, not taken from memory.
//
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/mx_probe.hip -o tools/bin/mx_probe && tools/bin/mx_probe
//
// Hypothesis tested (exit code 0 = all of it holds):
//   A operand: lane l holds row i = l & 15, K block g = l >> 4: elements k = 32 g + j, j = 0..31, as 32 nibbles in 4 VGPRs
//              (VGPR j / 8, nibble j % 8, nibble 0 = bits 3:0); B operand: lane l holds column n = l & 15, same K mapping;
//   D: lane l holds column n = l & 15 of rows i = 4 (l >> 4) + r, r = 0..3 (the 16-bit 16x16 map);
//   value of a nibble: e2m1 = {0, .5, 1, 1.5, 2, 3, 4, 6} x sign(bit 3);
//   scales: byte `opsel` of the scale VGPR of lane l is an e8m0 exponent: factor 2^(byte - 127) for that lane's 32 elements.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) {                                            \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
      exit(2);                                                         \
    }                                                                  \
  } while (0)

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

// one MFMA: A, B as 4 dwords per lane (fp4), scales as one dword per lane, opsel immediate
template <int OPA, int OPB>
__global__ void one_mfma(const i32x4* __restrict__ a, const i32x4* __restrict__ b, const int* __restrict__ sa, const int* __restrict__ sb,
                         f32x4* __restrict__ d) {
  const int l = threadIdx.x;
  const i32x4 av = a[l], bv = b[l];
  const i32x8 a8 = {av[0], av[1], av[2], av[3], 0, 0, 0, 0}, b8 = {bv[0], bv[1], bv[2], bv[3], 0, 0, 0, 0};
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c, 4, 4, OPA, sa[l], OPB, sb[l]);
  d[l] = c;
}

// timing: NACC independent accumulators, REP rounds; MODE 0 = 16-bit MFMA only, 1 = fp4 MX only, 2 = four 16-bit + one MX per
// accumulator round (the mix of the GEMM: K = 128 of hi next to K = 128 of lo)
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(int rep, float* out, long long* clk) {
  constexpr int NACC = 16;
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int l = threadIdx.x;
  bf16x8 ha = {(short)(0x3c00 + l), 1, 2, 3, 4, 5, 6, 7}, hb = {(short)(0x3c00 + 2 * l), 3, 1, 4, 1, 5, 9, 2};
  i32x8 a8 = {l * 0x01010101, 0x12345678, 0x23456701, 0x11111111, 0, 0, 0, 0}, b8 = {0x01234567, l, 0x22222222, 0x76543210, 0, 0, 0, 0};
  const int s = 0x7f7f7f7f;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rep; ++r) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[i], 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int q = 0; q < (MODE == 1 ? 4 : 1); ++q)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 4, 4, 0, s, 0, s);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + l] = t;
  if (l == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

// v_cvt_scalef32_pk_fp4_f32: two f32 -> two e2m1 codes in byte `sel` of the destination. Hypothesis: code(x / scale), round to nearest,
// ties to the even mantissa, saturating at 6; first source in bits 3:0
__global__ void cvt_kernel(const float* __restrict__ x, float scale, unsigned* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned o = 0xffffffffu;
  o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, x[2 * i], x[2 * i + 1], scale, 2);
  out[i] = o;
}

static const float E2M1[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static unsigned sw_code(float x) {   // the library's quantiser (vt_mx4.hip mx4_code) on the host
  const float a = std::fmin(std::fabs(x), 6.0f);
  const float c = a < 2.0f ? std::nearbyint(a * 2.0f) : (a < 4.0f ? std::nearbyint(a) + 2.0f : std::nearbyint(a * 0.5f) + 4.0f);
  unsigned u;
  memcpy(&u, &x, 4);
  return (unsigned)c | ((u >> 28) & 8u);
}
static float fp4_val(int nib) { return (nib & 8) ? -E2M1[nib & 7] : E2M1[nib & 7]; }

int main() {
  std::mt19937 rng(7);
  std::vector<int> A(16 * 128), B(16 * 128);   // nibble codes: A[i][k], B[n][k]
  for (auto& v : A) v = rng() & 15;
  for (auto& v : B) v = rng() & 15;
  std::vector<uint32_t> ha(64 * 4, 0), hb(64 * 4, 0);
  std::vector<int> hsa(64), hsb(64);
  for (int l = 0; l < 64; ++l) {
    const int i = l & 15, g = l >> 4;
    for (int j = 0; j < 32; ++j) {
      ha[l * 4 + j / 8] |= (uint32_t)A[i * 128 + 32 * g + j] << (4 * (j % 8));
      hb[l * 4 + j / 8] |= (uint32_t)B[i * 128 + 32 * g + j] << (4 * (j % 8));
    }
    // four different per-(row, block) exponents, one per byte: opsel = b must pick byte b
    hsa[l] = hsb[l] = 0;
    for (int b = 0; b < 4; ++b) {
      hsa[l] |= (118 + (int)(rng() % 16)) << (8 * b);
      hsb[l] |= (119 + (int)(rng() % 16)) << (8 * b);
    }
  }
  uint32_t *da, *db;
  int *dsa, *dsb;
  f32x4* dd;
  CK(hipMalloc(&da, 64 * 16));
  CK(hipMalloc(&db, 64 * 16));
  CK(hipMalloc(&dsa, 64 * 4));
  CK(hipMalloc(&dsb, 64 * 4));
  CK(hipMalloc(&dd, 64 * 16));
  CK(hipMemcpy(da, ha.data(), 64 * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), 64 * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, hsa.data(), 64 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsb, hsb.data(), 64 * 4, hipMemcpyHostToDevice));
  int bad_total = 0;
  for (int sel = 0; sel < 4; ++sel) {
    if (sel == 0) one_mfma<0, 0><<<1, 64>>>((const i32x4*)da, (const i32x4*)db, dsa, dsb, dd);
    if (sel == 1) one_mfma<1, 1><<<1, 64>>>((const i32x4*)da, (const i32x4*)db, dsa, dsb, dd);
    if (sel == 2) one_mfma<2, 2><<<1, 64>>>((const i32x4*)da, (const i32x4*)db, dsa, dsb, dd);
    if (sel == 3) one_mfma<3, 3><<<1, 64>>>((const i32x4*)da, (const i32x4*)db, dsa, dsb, dd);
    CK(hipDeviceSynchronize());
    std::vector<float> hd(64 * 4);
    CK(hipMemcpy(hd.data(), dd, 64 * 16, hipMemcpyDeviceToHost));
    int bad = 0;
    double maxrel = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int n = l & 15, i = 4 * (l >> 4) + r;   // D[i][n] = sum_k A[i][k] B[n][k]
        double ref = 0;
        for (int g = 0; g < 4; ++g) {
          double part = 0;
          for (int j = 0; j < 32; ++j) part += (double)fp4_val(A[i * 128 + 32 * g + j]) * fp4_val(B[n * 128 + 32 * g + j]);
          const int la = i + 16 * g, lb = n + 16 * g;
          const int ea = (hsa[la] >> (8 * sel)) & 255, eb = (hsb[lb] >> (8 * sel)) & 255;
          ref += part * std::ldexp(1.0, ea - 127) * std::ldexp(1.0, eb - 127);
        }
        const double got = hd[l * 4 + r];
        const double rel = std::fabs(got - ref) / (std::fabs(ref) + 1e-30);
        if (rel > maxrel) maxrel = rel;
        if (rel > 1e-6) {
          if (bad < 6) printf("  mismatch opsel=%d lane %d r %d: got %.9g want %.9g\n", sel, l, r, got, ref);
          ++bad;
        }
      }
    printf("{\"probe\": \"mx_fp4_16x16x128\", \"opsel\": %d, \"mismatches\": %d, \"max_rel\": %.3g}\n", sel, bad, maxrel);
    bad_total += bad;
  }
  {  // the conversion instruction against the software quantiser: a grid of values (ties included), random values, three scales
    std::vector<float> xs;
    for (int i = -60; i <= 60; ++i) xs.push_back(i * 0.125f);            // every tie of the e2m1 grid and the saturation edge
    for (int i = 0; i < 4000; ++i) xs.push_back(((int)(rng() % 200001) - 100000) * 7e-5f);
    xs.push_back(1e-9f); xs.push_back(-1e-9f); xs.push_back(100.f); xs.push_back(-100.f);
    if (xs.size() & 1) xs.push_back(0.f);
    const int n = (int)xs.size() / 2;
    float* dx;
    unsigned* dout;
    CK(hipMalloc(&dx, xs.size() * 4));
    CK(hipMalloc(&dout, n * 4));
    for (float scale : {1.0f, 0.25f, 8.0f}) {
      std::vector<float> scaled(xs);
      for (auto& v : scaled) v *= scale;                                  // so that x / scale is the grid value again
      CK(hipMemcpy(dx, scaled.data(), scaled.size() * 4, hipMemcpyHostToDevice));
      cvt_kernel<<<(n + 255) / 256, 256>>>(dx, scale, dout, n);
      CK(hipDeviceSynchronize());
      std::vector<unsigned> ho(n);
      CK(hipMemcpy(ho.data(), dout, n * 4, hipMemcpyDeviceToHost));
      int bad = 0, bad_other_bytes = 0;
      for (int i = 0; i < n; ++i) {
        const unsigned want = sw_code(xs[2 * i]) | (sw_code(xs[2 * i + 1]) << 4);
        const unsigned got = (ho[i] >> 16) & 255u;
        if ((ho[i] & 0xff00ffffu) != 0xff00ffffu) ++bad_other_bytes;
        if (got != want) {
          if (bad < 8) printf("  cvt mismatch scale %g: (%g, %g) -> got %02x want %02x\n", scale, xs[2 * i], xs[2 * i + 1], got, want);
          ++bad;
        }
      }
      printf("{\"probe\": \"cvt_scalef32_pk_fp4_f32\", \"scale\": %g, \"pairs\": %d, \"mismatches\": %d, \"other_bytes_touched\": %d}\n", scale, n, bad,
             bad_other_bytes);
      bad_total += bad + bad_other_bytes;
    }
  }
  // rates: one wave per SIMD on every CU
  float* dout;
  long long* dclk;
  CK(hipMalloc(&dout, 256 * 256 * 4));
  CK(hipMalloc(&dclk, 8));
  const int rep = 4000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double us[3];
  long long clk[3];
  for (int mode = 0; mode < 3; ++mode) {
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipEventRecord(e0));
      if (mode == 0) rate_kernel<0><<<256, 256>>>(rep, dout, dclk);
      if (mode == 1) rate_kernel<1><<<256, 256>>>(rep, dout, dclk);
      if (mode == 2) rate_kernel<2><<<256, 256>>>(rep, dout, dclk);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      us[mode] = ms * 1e3;
    }
    CK(hipMemcpy(&clk[mode], dclk, 8, hipMemcpyDeviceToHost));
  }
  const double n16 = 64.0 * rep, nmx1 = 64.0 * rep, nmx2 = 16.0 * rep;
  printf("{\"probe\": \"rate\", \"mfma16_only\": {\"us\": %.1f, \"cycles_per_mfma\": %.2f, \"pflops\": %.3f}, "
         "\"mx4_only\": {\"us\": %.1f, \"cycles_per_mfma\": %.2f, \"pflops\": %.3f}, "
         "\"mix_4to1\": {\"us\": %.1f, \"cycles_per_round_of_5\": %.2f, \"vs_mfma16_only\": %.3f}}\n",
         us[0], clk[0] / n16, 1024.0 * n16 * 2 * 16 * 16 * 32 / us[0] * 1e-9,
         us[1], clk[1] / nmx1, 1024.0 * nmx1 * 2 * 16 * 16 * 128 / us[1] * 1e-9,
         us[2], clk[2] / nmx2, us[2] / us[0]);
  return bad_total ? 1 : 0;
}
