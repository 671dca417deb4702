// gemm_ab -- interleaved, sustained A/B of vt_gemm_bf16 tile configurations on one MI355X, without a Python/torch start-up.
//
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/gemm_ab.cpp -Iinclude -Lvitron_amd -lvitron_hip -Wl,-rpath,'$ORIGIN/../../vitron_amd' -o tools/bin/gemm_ab
//   tools/bin/gemm_ab  M,N,K,epi[;M,N,K,epi...]  cfgA,cfgB,...  [seconds per window = 0.6]  [rounds = 3]
//
// Every window launches one configuration back to back for `seconds` (long enough for the power limit to settle: the sustained
// clock of this part is data- and history-dependent), windows of the configurations alternate, the median over
// the rounds is reported. Operands: A ~ N(0,1), W ~ N(0,0.02^2) in bf16 (what the decoder sees). The first launch of every
// configuration is compared element by element with the first configuration's output (max |diff|), so a variant that is fast
// because it is wrong shows up here. Measurement helper, not part of the product path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "vitron_hip.h"

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                        \
      exit(2);                                                                       \
    }                                                                                \
  } while (0)

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static std::vector<std::string> split(const std::string& s, char c) {
  std::vector<std::string> out;
  size_t p = 0;
  while (true) {
    size_t q = s.find(c, p);
    out.push_back(s.substr(p, q == std::string::npos ? q : q - p));
    if (q == std::string::npos) break;
    p = q + 1;
  }
  return out;
}

static void fill(uint16_t* dev, size_t n, float std_, uint32_t seed) {
  // a 1M-element random block repeated with a per-repeat sign/offset twist: cheap on the host, still "random data" to the MFMA
  const size_t blk = 1 << 20;
  std::vector<uint16_t> h(std::min(n, blk));
  if (getenv("GEMM_AB_DATA") && !strcmp(getenv("GEMM_AB_DATA"), "zeros")) {   // no data-dependent power: the schedule's own ceiling
    CK(hipMemset(dev, 0, n * 2));
    return;
  }
  std::mt19937 g(seed);
  std::normal_distribution<float> d(0.f, std_);
  for (auto& v : h) v = f2bf(d(g));
  for (size_t off = 0; off < n; off += blk) {
    const size_t c = std::min(blk, n - off);
    const size_t rot = (off / blk * 7919) % h.size();
    std::rotate(h.begin(), h.begin() + (rot % h.size()), h.end());
    CK(hipMemcpy(dev + off, h.data(), c * 2, hipMemcpyHostToDevice));
  }
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: gemm_ab M,N,K,epi[;...] cfg,cfg,... [seconds] [rounds]\n");
    return 1;
  }
  const double seconds = argc > 3 ? atof(argv[3]) : 0.6;
  const int rounds = argc > 4 ? atoi(argv[4]) : 3;
  std::vector<int> cfgs;
  for (auto& c : split(argv[2], ',')) cfgs.push_back(atoi(c.c_str()));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (auto& shp : split(argv[1], ';')) {
    auto f = split(shp, ',');
    if (f.size() < 4) continue;
    const int M = atoi(f[0].c_str()), N = atoi(f[1].c_str()), K = atoi(f[2].c_str()), epi = atoi(f[3].c_str());
    const bool f32out = epi == VT_EPI_F32 || epi == VT_EPI_F32_RESID;
    const int ldc = epi == VT_EPI_SWIGLU_BF16 ? N / 2 : N;
    const size_t cbytes = (size_t)M * ldc * (f32out ? 4 : 2);
    const int nrot = getenv("GEMM_AB_ROTATE") ? std::max(1, atoi(getenv("GEMM_AB_ROTATE"))) : 1;   // weight copies visited round-robin:
    uint16_t *A, *W;                                                                                // > 1 keeps W out of the Infinity Cache
    void *C, *Cref;
    CK(hipMalloc(&A, (size_t)M * K * 2));
    CK(hipMalloc(&W, (size_t)N * K * 2 * nrot));
    CK(hipMalloc(&C, cbytes));
    CK(hipMalloc(&Cref, cbytes));
    fill(A, (size_t)M * K, 1.0f, 1);
    for (int r = 0; r < nrot; ++r) fill(W + (size_t)r * N * K, (size_t)N * K, 0.02f, 2);   // identical copies: same result whichever is used
    std::vector<std::vector<double>> us(cfgs.size());
    std::vector<double> maxdiff(cfgs.size(), 0.0);
    std::vector<char> hc(cbytes), hr(cbytes);
    for (size_t ci = 0; ci < cfgs.size(); ++ci) {
      CK(hipMemsetAsync(C, 0, cbytes, st));
      int rc = vt_gemm_bf16(A, K, W, K, C, ldc, nullptr, M, N, K, epi, cfgs[ci], nullptr, st);
      if (rc != 0) {
        char buf[512];
        vt_last_error(buf, sizeof buf);
        fprintf(stderr, "cfg %d: %s\n", cfgs[ci], buf);
        maxdiff[ci] = -1;
        continue;
      }
      CK(hipStreamSynchronize(st));
      if (ci == 0) {
        CK(hipMemcpy(Cref, C, cbytes, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(hr.data(), C, cbytes, hipMemcpyDeviceToHost));
      } else {
        CK(hipMemcpy(hc.data(), C, cbytes, hipMemcpyDeviceToHost));
        double md = 0;
        const size_t n = (size_t)M * ldc;
        for (size_t i = 0; i < n; ++i) {
          const double a = f32out ? ((float*)hc.data())[i] : bf2f(((uint16_t*)hc.data())[i]);
          const double b = f32out ? ((float*)hr.data())[i] : bf2f(((uint16_t*)hr.data())[i]);
          const double dd = std::fabs(a - b);
          if (!(dd <= md)) md = dd;   // NaN propagates
        }
        maxdiff[ci] = md;
      }
    }
    for (int r = 0; r < rounds; ++r)
      for (size_t ci = 0; ci < cfgs.size(); ++ci) {
        if (maxdiff[ci] < 0) continue;
        if (epi == VT_EPI_F32_RESID) CK(hipMemsetAsync(C, 0, cbytes, st));
        long n = 0;
        CK(hipStreamSynchronize(st));
        const auto t0 = std::chrono::steady_clock::now();
        double el = 0;
        while (el < seconds) {
          for (int i = 0; i < 20; ++i) vt_gemm_bf16(A, K, W + (size_t)((n + i) % nrot) * N * K, K, C, ldc, nullptr, M, N, K, epi, cfgs[ci], nullptr, st);
          CK(hipStreamSynchronize(st));
          n += 20;
          el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        us[ci].push_back(el / n * 1e6);
      }
    printf("{\"shape\": [%d, %d, %d], \"epi\": %d, \"results\": {", M, N, K, epi);
    for (size_t ci = 0; ci < cfgs.size(); ++ci) {
      if (us[ci].empty()) {
        printf("%s\"%d\": null", ci ? ", " : "", cfgs[ci]);
        continue;
      }
      std::sort(us[ci].begin(), us[ci].end());
      const double med = us[ci][us[ci].size() / 2];
      printf("%s\"%d\": {\"us\": %.1f, \"tflops\": %.1f, \"min_us\": %.1f, \"maxdiff_vs_first\": %.3g}", ci ? ", " : "", cfgs[ci], med,
             2.0 * M * N * K / med / 1e6, us[ci][0], maxdiff[ci]);
    }
    printf("}}\n");
    fflush(stdout);
    CK(hipFree(A));
    CK(hipFree(W));
    CK(hipFree(C));
    CK(hipFree(Cref));
  }
  return 0;
}
