// vt_llama.hip -- small kernels around the decoder: embedding gather + multimodal splice, greedy argmax.
//
// embed_splice replaces the tensor surgery of prepare_inputs_labels_for_multimodal in the reference
// (vitron/model/llava_arch.py:306-398 region branch, :479-558 plain branch): embed_tokens gather for text
// ids, visual feature blocks at every -200 sentinel, one region feature row at every -300 sentinel,
// truncation and zero padding. The integer plan (which source row feeds which output row) is computed on
// the host by vitron_amd.model.llava_arch.build_splice_plan, bit-exactly following the reference's list
// logic; this kernel only moves rows.
#include "vt_common.h"
#include "vt_kernels.h"

namespace {

// plan[r] = {kind, index}: kind 0 = token id into tok_table, 1 = row of vis, 2 = row of reg, 3 = zeros
__global__ __launch_bounds__(256) void embed_splice_kernel(const bf16_t* __restrict__ tok_table,
                                                           const bf16_t* __restrict__ vis,
                                                           const bf16_t* __restrict__ reg,
                                                           const int* __restrict__ plan, int rows, int H,
                                                           bf16_t* __restrict__ out, int vocab, int vis_rows, int reg_rows) {
  const int chunks = H >> 3;  // 16-B chunks per row
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)rows * chunks;
       i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / chunks), c = (int)(i % chunks);
    const int kind = plan[2 * r], idx = plan[2 * r + 1];
    u32x4 v = {0u, 0u, 0u, 0u};
    // an index outside its table (a stray sentinel, an id >= the embedding rows) yields a zero row, never an out-of-bounds read;
    // the host mirror range-checks ids and raises like the reference's nn.Embedding would (vitron_amd/model/llava_arch.py)
    if (kind == 0 && (unsigned)idx < (unsigned)vocab) v = *(const u32x4*)(tok_table + (size_t)idx * H + c * 8);
    else if (kind == 1 && (unsigned)idx < (unsigned)vis_rows) v = *(const u32x4*)(vis + (size_t)idx * H + c * 8);
    else if (kind == 2 && (unsigned)idx < (unsigned)reg_rows) v = *(const u32x4*)(reg + (size_t)idx * H + c * 8);
    *(u32x4*)(out + (size_t)r * H + c * 8) = v;
  }
}

// decode-loop feedback, one block per sequence: thread 0 resolves the token this sequence emits (pad once finished), updates
// the finished flag against the EOS set and advances the sequence's device-side step metadata (kv_len, position); the block
// then copies the token's embedding row into the next step's input. Everything the next decoder pass needs is on the device
// after this launch, so the host can enqueue that pass before it has seen the token.
__global__ __launch_bounds__(256) void decode_feed_kernel(const bf16_t* __restrict__ tok_table, int H, int vocab,
                                                          const int* __restrict__ next_ids, int* __restrict__ finished,
                                                          const int* __restrict__ eos_ids, int n_eos, int pad_id,
                                                          int* __restrict__ tokens_out, bf16_t* __restrict__ x,
                                                          int* __restrict__ seq_desc, int* __restrict__ positions, int nseq) {
  __shared__ int tok_s;
  const int i = blockIdx.x;
  if (threadIdx.x == 0) {
    int fin = finished[i];
    const int t = fin ? pad_id : next_ids[i];
    for (int e = 0; e < n_eos; ++e) fin |= (t == eos_ids[e]);
    finished[i] = fin;
    tokens_out[i] = t;
    tokens_out[nseq + i] = fin;
    seq_desc[4 * i + 2] += 1;   // VtAttnSeq::kv_len
    positions[i] += 1;
    tok_s = min(max(t, 0), vocab - 1);
  }
  __syncthreads();
  const bf16_t* src = tok_table + (size_t)tok_s * H;
  bf16_t* dst = x + (size_t)i * H;
  for (int c = threadIdx.x; c < (H >> 3); c += 256) *(u32x4*)(dst + c * 8) = *(const u32x4*)(src + c * 8);
}

// first index of the maximum of each row (torch.argmax tie rule on CPU: lowest index). 1024 threads per row, 16-B loads,
// every request of the row in flight at once (a 32000-float row is 8 loads per thread).
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, int V, int ldl,
                                                      int* __restrict__ out) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const float* row = logits + (size_t)blockIdx.x * ldl;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  auto take = [&](float v, int i) {
    if (v > best || (v == best && i < bi)) {
      best = v;
      bi = i;
    }
  };
  const bool vec = (ldl % 4) == 0 && ((uintptr_t)logits % 16) == 0;
  const int V4 = vec ? (V >> 2) : 0;
#pragma unroll 8
  for (int i = threadIdx.x; i < V4; i += 1024) {
    const f32x4 v = *(const f32x4*)(row + 4 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) take(v[k], 4 * i + k);
  }
  for (int i = 4 * V4 + threadIdx.x; i < V; i += 1024) take(row[i], i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    take(ov, oi);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sv[wave] = best;
    si[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) take(sv[w], si[w]);
    out[blockIdx.x] = bi;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Temperature / top-k-free top-p sampling of one token per row, entirely on the device (no logits copy, no host sync):
//   p = softmax(logits / T);  keep the smallest set of highest-probability tokens whose mass reaches top_p -- exactly
//   transformers' TopPLogitsWarper rule (sort ascending, drop while cumulative mass <= 1 - top_p, always keep one) --
//   found by bisection on the probability threshold instead of a sort; then inverse-CDF sampling over the kept tokens in
//   index order with a counter-based uniform (splitmix64 of seed, step, row). One 1024-thread block per row.
// Replaces GenerationMixin.sample's warper + torch.multinomial (app.py:562-571 passes do_sample=True, temperature, top_p).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_1024(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < 16; ++i) t += sh[i];
  return t;
}
__device__ __forceinline__ float block_max_1024(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = sh[0];
  for (int i = 1; i < 16; ++i) t = fmaxf(t, sh[i]);
  return t;
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

// monotone map float -> uint32 (a < b  <=>  key(a) < key(b)); -inf maps below every finite value
__device__ __forceinline__ uint32_t order_key(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// TopKLogitsWarper (transformers 4.31: scores < topk(scores, k)[..., -1] are removed, ties with the k-th value stay): the
// k-th largest scaled logit is found exactly by bisection on the ordered bit pattern (32 counting passes), no sort.
__global__ __launch_bounds__(1024) void sample_top_p_kernel(const float* __restrict__ logits, int V, int ldl,
                                                            float inv_temp, int top_k, float top_p, uint64_t seed, uint64_t step,
                                                            int* __restrict__ out_ids, int* __restrict__ kept_count) {
  __shared__ float sh[16];
  __shared__ float sh_scan[16];
  const float* row = logits + (size_t)blockIdx.x * ldl;
  const int tid = threadIdx.x;
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, row[i] * inv_temp);
  mx = block_max_1024(mx, sh);
  float floor_logit = -INFINITY;   // scaled logits below this are outside the top-k set
  if (top_k > 0 && top_k < V) {
    uint64_t lo = 0, hi = 0x100000000ull;   // invariant: count(key >= lo) >= k, count(key >= hi) < k
    for (int it = 0; it < 32; ++it) {
      const uint64_t mid = lo + ((hi - lo) >> 1);
      float c = 0.f;
      for (int i = tid; i < V; i += 1024) c += ((uint64_t)order_key(row[i] * inv_temp) >= mid) ? 1.f : 0.f;
      c = block_sum_1024(c, sh);
      if (c >= (float)top_k) lo = mid; else hi = mid;
    }
    const uint32_t k32 = (uint32_t)lo;
    floor_logit = __uint_as_float((k32 & 0x80000000u) ? (k32 & 0x7fffffffu) : ~k32);
  }
#define VT_P_OF(i) ((row[i] * inv_temp >= floor_logit) ? __expf(row[i] * inv_temp - mx) : 0.f)
  float z = 0.f;
  for (int i = tid; i < V; i += 1024) z += VT_P_OF(i);
  z = block_sum_1024(z, sh);
  const float inv_z = 1.f / z;
  // bisection on the threshold t in (0, 1]: mass(t) = sum of p_i with p_i >= t is non-increasing in t.
  // keep-set of the warper = {i : mass of tokens strictly above p_i < top_p}  <=>  largest t with mass(t) >= top_p.
  float lo = 0.f, hi = 1.f;   // invariant: mass(lo) >= top_p, mass(hi) < top_p (or hi == 1 when a single token has p >= top_p)
  for (int it = 0; it < 30; ++it) {
    const float mid = 0.5f * (lo + hi);
    float m = 0.f;
    for (int i = tid; i < V; i += 1024) {
      const float p = VT_P_OF(i) * inv_z;
      if (p >= mid) m += p;
    }
    m = block_sum_1024(m, sh);
    if (m >= top_p) lo = mid; else hi = mid;
  }
  const float thr = (top_p >= 1.f) ? 0.f : fminf(lo, inv_z);   // inv_z = probability of the arg-max token: always kept
  float kept = 0.f, cnt = 0.f;
  for (int i = tid; i < V; i += 1024) {
    const float p = VT_P_OF(i) * inv_z;
    if (p >= thr && row[i] * inv_temp >= floor_logit) {      // (thr == 0 at top_p >= 1: the top-k filter still applies)
      kept += p;
      cnt += 1.f;
    }
  }
  kept = block_sum_1024(kept, sh);
  cnt = block_sum_1024(cnt, sh);
  if (tid == 0 && kept_count) kept_count[blockIdx.x] = (int)cnt;
  // inverse CDF over the kept tokens in index order: thread t owns the contiguous index range [t*chunk, (t+1)*chunk)
  const uint64_t r = splitmix64(seed ^ splitmix64(step * 0x632be59bd9b4e019ull + blockIdx.x));
  const float u = (float)((r >> 40) + 0.5) * (1.0f / 16777216.0f) * kept;
  const int chunk = (V + 1023) / 1024;
  const int i0 = tid * chunk, i1 = min(V, i0 + chunk);
  float mine = 0.f;
  for (int i = i0; i < i1; ++i) {
    const float p = VT_P_OF(i) * inv_z;
    if (p >= thr && row[i] * inv_temp >= floor_logit) mine += p;
  }
  // exclusive prefix over threads: wave scan + scan of the 16 wave totals
  float incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float v = __shfl_up(incl, o, 64);
    if ((tid & 63) >= o) incl += v;
  }
  __syncthreads();
  if ((tid & 63) == 63) sh_scan[tid >> 6] = incl;
  __syncthreads();
  float base = 0.f;
  for (int w = 0; w < (tid >> 6); ++w) base += sh_scan[w];
  const float excl = base + incl - mine;
  // the owner of u is the thread with excl <= u < excl + mine; fall back to the last kept token against rounding
  __shared__ int chosen;
  if (tid == 0) chosen = -1;
  __syncthreads();
  if (mine > 0.f && u >= excl && u < excl + mine) {
    float c = excl;
    int pick = -1;
    for (int i = i0; i < i1; ++i) {
      const float p = VT_P_OF(i) * inv_z;
      if (p >= thr && row[i] * inv_temp >= floor_logit) {
        pick = i;
        c += p;
        if (u < c) break;
      }
    }
    atomicMax(&chosen, pick);
  }
  __syncthreads();
  if (chosen < 0 && mine > 0.f) {  // u landed on a rounding seam: take the last kept token of the highest owning range
    int last = -1;
    for (int i = i0; i < i1; ++i)
      if (VT_P_OF(i) * inv_z >= thr && row[i] * inv_temp >= floor_logit) last = i;
    atomicMax(&chosen, last);
  }
  __syncthreads();
  if (tid == 0) out_ids[blockIdx.x] = chosen;
}
#undef VT_P_OF

// Same rule with the row held in registers (V <= 32 * 1024): every thread owns 32 consecutive logits, computes their
// exponentials ONCE, and the 30 bisection passes, the kept-mass pass and the inverse-CDF walk never touch memory again.
// 165 us -> ~20 us per call at V = 32000 (tools/sampler_bench.py).
__global__ __launch_bounds__(1024) void sample_top_p_reg_kernel(const float* __restrict__ logits, int V, int ldl,
                                                                float inv_temp, int top_k, float top_p, uint64_t seed, uint64_t step,
                                                                int* __restrict__ out_ids, int* __restrict__ kept_count) {
  constexpr int NPT = 32;
  __shared__ float sh[2][16];
  __shared__ float sh_scan[16];
  __shared__ int chosen;
  const float* row = logits + (size_t)blockIdx.x * ldl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = tid * NPT;
  float p[NPT];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NPT; j += 4) {
    f32x4 v = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (i0 + j + 3 < V) {
      v = *(const f32x4*)(row + i0 + j);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i0 + j + k < V) v[k] = row[i0 + j + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      p[j + k] = v[k] * inv_temp;
      mx = fmaxf(mx, p[j + k]);
    }
  }
  int buf = 0;
  auto block_sum = [&](float v) -> float {   // one barrier per call: two alternating LDS rows
    v = wave_sum(v);
    if (lane == 0) sh[buf][wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[buf][i];
    buf ^= 1;
    return t;
  };
  {
    mx = wave_max(mx);
    if (lane == 0) sh[buf][wave] = mx;
    __syncthreads();
    float t = sh[buf][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) t = fmaxf(t, sh[buf][i]);
    mx = t;
    buf ^= 1;
  }
  uint32_t valid = 0xffffffffu;   // bit j: slot j is inside the top-k set (all slots when the filter is off)
  if (top_k > 0 && top_k < V) {   // TopKLogitsWarper: everything below the k-th largest scaled logit leaves the distribution
    uint64_t lo = 0, hi = 0x100000000ull;
    for (int it = 0; it < 32; ++it) {
      const uint64_t mid = lo + ((hi - lo) >> 1);
      float c = 0.f;
#pragma unroll
      for (int j = 0; j < NPT; ++j) c += ((uint64_t)order_key(p[j]) >= mid) ? 1.f : 0.f;
      c = block_sum(c);
      if (c >= (float)top_k) lo = mid; else hi = mid;
    }
    valid = 0u;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const bool in = (uint64_t)order_key(p[j]) >= lo;
      valid |= in ? (1u << j) : 0u;
      p[j] = in ? p[j] : -INFINITY;
    }
  }
  float zl = 0.f;
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    p[j] = __expf(p[j] - mx);     // exp(-inf) = 0 for the slots past V and outside the top-k set
    zl += p[j];
  }
  const float z = block_sum(zl);
  const float inv_z = 1.f / z;
#pragma unroll
  for (int j = 0; j < NPT; ++j) p[j] *= inv_z;
  float lo = 0.f, hi = 1.f;   // invariant: mass(lo) >= top_p, mass(hi) < top_p (or hi == 1 when one token has p >= top_p)
  for (int it = 0; it < 30; ++it) {
    const float mid = 0.5f * (lo + hi);
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < NPT; ++j) m += (p[j] >= mid) ? p[j] : 0.f;
    m = block_sum(m);
    if (m >= top_p) lo = mid; else hi = mid;
  }
  const float thr = (top_p >= 1.f) ? 0.f : fminf(lo, inv_z);   // inv_z = probability of the arg-max token: always kept
  float mine = 0.f, cnt = 0.f;
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const bool keep = p[j] >= thr && i0 + j < V && ((valid >> j) & 1u);
    p[j] = keep ? p[j] : 0.f;       // from here on p holds kept probabilities only
    mine += p[j];
    cnt += keep ? 1.f : 0.f;
  }
  const float kept = block_sum(mine);
  cnt = block_sum(cnt);
  if (tid == 0) {
    if (kept_count) kept_count[blockIdx.x] = (int)cnt;
    chosen = -1;
  }
  const uint64_t r = splitmix64(seed ^ splitmix64(step * 0x632be59bd9b4e019ull + blockIdx.x));
  const float u = (float)((r >> 40) + 0.5) * (1.0f / 16777216.0f) * kept;
  // exclusive prefix over threads: wave scan + scan of the 16 wave totals
  float incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) sh_scan[wave] = incl;
  __syncthreads();
  float base = 0.f;
  for (int w = 0; w < wave; ++w) base += sh_scan[w];
  const float excl = base + incl - mine;
  if (mine > 0.f && u >= excl && u < excl + mine) {
    float c = excl;
    int pick = -1;
    bool done = false;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      if (!done && p[j] > 0.f) {
        pick = i0 + j;
        c += p[j];
        done = u < c;
      }
    }
    atomicMax(&chosen, pick);
  }
  __syncthreads();
  if (chosen < 0 && mine > 0.f) {  // u landed on a rounding seam: take the last kept token of the highest owning range
    int last = -1;
#pragma unroll
    for (int j = 0; j < NPT; ++j)
      if (p[j] > 0.f) last = i0 + j;
    atomicMax(&chosen, last);
  }
  __syncthreads();
  if (tid == 0) out_ids[blockIdx.x] = chosen;
}

// ---------------------------------------------------------------------------------------------------------------
// Shifted-token cross entropy of LlamaForCausalLM.forward(labels=...) (ignore_index rows skipped, mean over the rest):
//   nll[r] = logsumexp(logits[r]) - logits[r][label[r]]    one 1024-thread block per row, then one block averages.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void cross_entropy_rows_kernel(const float* __restrict__ logits, int V, int ldl,
                                                                  const int* __restrict__ labels, int ignore_index,
                                                                  float* __restrict__ nll) {
  __shared__ float sh[16];
  const int r = blockIdx.x;
  const int lab = labels[r];
  if (lab == ignore_index || lab < 0 || lab >= V) {     // (an out-of-range label cannot index the row: it is skipped like ignore_index)
    if (threadIdx.x == 0) nll[r] = -1.f;                // marker: not counted
    return;
  }
  const float* row = logits + (size_t)r * ldl;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < V; i += 1024) mx = fmaxf(mx, row[i]);
  mx = block_max_1024(mx, sh);
  float z = 0.f;
  for (int i = threadIdx.x; i < V; i += 1024) z += expf(row[i] - mx);
  z = block_sum_1024(z, sh);
  if (threadIdx.x == 0) nll[r] = fmaxf(logf(z) + mx - row[lab], 0.f);
}
__global__ __launch_bounds__(1024) void cross_entropy_mean_kernel(const float* __restrict__ nll, int rows, float* __restrict__ loss) {
  __shared__ float sh[16];
  float s = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < rows; i += 1024) {
    const float v = nll[i];
    if (v >= 0.f) {
      s += v;
      c += 1.f;
    }
  }
  s = block_sum_1024(s, sh);
  c = block_sum_1024(c, sh);
  if (threadIdx.x == 0) loss[0] = s / c;                // no valid row: 0/0 = NaN, as torch's mean over nothing
}

}  // namespace

int vt_cross_entropy_launch(const float* logits, int rows, int V, int ldl, const int* labels, int ignore_index, float* row_nll,
                            float* loss, hipStream_t s) {
  VT_REQUIRE(logits && labels && row_nll && loss && rows > 0 && V > 0, "vt_cross_entropy: bad arguments");
  hipLaunchKernelGGL(cross_entropy_rows_kernel, dim3(rows), dim3(1024), 0, s, logits, V, ldl, labels, ignore_index, row_nll);
  hipLaunchKernelGGL(cross_entropy_mean_kernel, dim3(1), dim3(1024), 0, s, row_nll, rows, loss);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_sample_top_p_launch(const float* logits, int rows, int V, int ldl, float temperature, int top_k, float top_p, uint64_t seed,
                           uint64_t step, int* out_ids, int* kept_count, hipStream_t s) {
  VT_REQUIRE(logits && out_ids && rows > 0 && V > 0, "vt_sample_top_p: bad arguments");
  VT_REQUIRE(temperature > 0.f && top_p > 0.f && top_k >= 0, "vt_sample_top_p: temperature and top_p must be > 0, top_k >= 0");
  if (V <= 32 * 1024 && (ldl % 4) == 0 && ((uintptr_t)logits % 16) == 0)
    hipLaunchKernelGGL(sample_top_p_reg_kernel, dim3(rows), dim3(1024), 0, s, logits, V, ldl, 1.0f / temperature, top_k, top_p, seed,
                       step, out_ids, kept_count);
  else
    hipLaunchKernelGGL(sample_top_p_kernel, dim3(rows), dim3(1024), 0, s, logits, V, ldl, 1.0f / temperature, top_k, top_p, seed,
                       step, out_ids, kept_count);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_embed_splice_launch(const bf16_t* tok_table, int vocab, const bf16_t* vis, int vis_rows, const bf16_t* reg, int reg_rows,
                           const int* plan, int rows, int H, bf16_t* out, hipStream_t s) {
  VT_REQUIRE(tok_table && plan && out, "vt_embed_splice: null pointer");
  VT_REQUIRE(vocab > 0 && vis_rows >= 0 && reg_rows >= 0 && (vis || vis_rows == 0) && (reg || reg_rows == 0),
             "vt_embed_splice: table sizes (vocab=%d vis_rows=%d reg_rows=%d)", vocab, vis_rows, reg_rows);
  VT_REQUIRE(rows > 0 && H % 8 == 0, "vt_embed_splice: rows=%d H=%d (H must be a multiple of 8)", rows, H);
  const long total = (long)rows * (H / 8);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(embed_splice_kernel, dim3(blocks), dim3(256), 0, s, tok_table, vis, reg, plan, rows, H, out, vocab, vis_rows, reg_rows);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_decode_feed_launch(const bf16_t* tok_table, int H, int vocab, const int* next_ids, int* finished, const int* eos_ids,
                          int n_eos, int pad_id, int* tokens_out, bf16_t* x, int* seq_desc, int* positions, int nseq,
                          hipStream_t s) {
  VT_REQUIRE(tok_table && next_ids && finished && tokens_out && x && seq_desc && positions, "vt_decode_feed: null pointer");
  VT_REQUIRE(nseq > 0 && vocab > 0 && H > 0 && H % 8 == 0, "vt_decode_feed: nseq=%d vocab=%d H=%d (H must be a multiple of 8)",
             nseq, vocab, H);
  VT_REQUIRE(n_eos == 0 || eos_ids, "vt_decode_feed: n_eos=%d without eos_ids", n_eos);
  hipLaunchKernelGGL(decode_feed_kernel, dim3(nseq), dim3(256), 0, s, tok_table, H, vocab, next_ids, finished, eos_ids,
                     n_eos < 0 ? 0 : n_eos, pad_id, tokens_out, x, seq_desc, positions, nseq);
  VT_LAUNCH_CHECK();
  return VT_OK;
}

int vt_argmax_launch(const float* logits, int rows, int V, int ldl, int* out_ids, hipStream_t s) {
  VT_REQUIRE(logits && out_ids && rows > 0 && V > 0, "vt_argmax: bad arguments");
  hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(1024), 0, s, logits, V, ldl, out_ids);
  VT_LAUNCH_CHECK();
  return VT_OK;
}
