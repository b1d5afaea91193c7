// Temperature / top-p (nucleus) sampling over a row of bf16 logits, one launch, one workgroup per sequence.
//
// Replaces the per-token sampling of `vllm.SamplingParams(temperature=0.6, top_p=0.9, ...)` inside `LLM.generate`
// (reference thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py:817-823, thinkdiff/models/mllama_vllm_generate_1.py:398-405)
// -- softmax(logits / T), keep the smallest set of most likely tokens whose mass reaches top_p, renormalise, draw one -- which
// a framework expresses as softmax + sort + cumsum + mask + multinomial over 152 064 logits (5 launches, a full sort, a host
// round trip) per generated token.
//
// No sort: bf16 logits have 65 536 distinct values, so the nucleus boundary is found by a two-level radix select on the
// order-preserving 16-bit key of the logit, with MASS histograms instead of count histograms:
//   pass 0  row maximum;
//   pass 1  mass per high key byte (256 bins) -> the bin in which the cumulative mass from the top reaches top_p * Z;
//   pass 2  mass and count per low key byte inside that bin -> the boundary key tau, the mass above it, and how many of the
//           tokens tied at tau are still inside the nucleus (a token is kept iff the mass in front of it is < top_p * Z, the
//           rule of the sort-based form);
//   pass 3  one uniform draw r in [0, kept mass) and a two-level prefix sum over the kept tokens to find the token r falls on.
// The row (304 KB for Qwen2-VL-7B) is re-read from L2 in each pass with 16-byte loads; nothing is written but the token id.
// All masses are 2^-40 fixed point in 64-bit integers: sums do not depend on the order of the LDS atomics, so a (seed,
// offset, row) triple always yields the same token.  HBM/L2-bound, ~4 reads of the row per token.
#include "td_common.h"
#include "td_kernels.h"

namespace {

constexpr int NT = 1024;            // threads per workgroup
constexpr int NW = NT / 64;         // waves
typedef unsigned long long u64;

// larger logit <-> larger key; NaN sorts below everything
__device__ __forceinline__ unsigned key16(unsigned b) {
  if ((b & 0x7fffu) > 0x7f80u) b = 0xff80u;                       // NaN -> -inf
  return (b & 0x8000u) ? (~b & 0xffffu) : (b | 0x8000u);
}
__device__ __forceinline__ float unkey16(unsigned k) {
  const unsigned b = (k & 0x8000u) ? (k & 0x7fffu) : (~k & 0xffffu);
  return __builtin_bit_cast(float, b << 16);
}
// un-normalised probability of a key as 2^-40 fixed point: exp((x - xmax) / T) in [0, 1]
__device__ __forceinline__ u64 mass_of(unsigned k, float xmax, float c) {
  const float e = __builtin_amdgcn_exp2f((unkey16(k) - xmax) * c);
  return (u64)(e * 1099511627776.0f);
}
__device__ __forceinline__ u64 splitmix64(u64 z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// exclusive prefix sum of one value per thread over the workgroup (thread order); total in *tot.  `scratch`: NW + 1 slots.
template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* scratch, T* tot) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  T inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const T up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) scratch[w] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    T run = 0;
    for (int i = 0; i < NW; ++i) { const T t = scratch[i]; scratch[i] = run; run += t; }
    scratch[NW] = run;
  }
  __syncthreads();
  const T base = scratch[w];
  *tot = scratch[NW];
  __syncthreads();
  return base + inc - v;
}

// Among 256 bins (mass in bins[], thread t < 256 owns bin t): the bin b, scanning from the top, in which the running mass first
// reaches `target` given `base` mass already in front of bin 255.  Writes {b, mass in front of b} for the one bin that matches.
__device__ __forceinline__ void find_crossing(const u64* bins, u64 base, u64 target, u64* scratch, int* out_bin, u64* out_above) {
  // suffix sum = prefix sum in reversed bin order; threads >= 256 contribute zeros
  const int t = threadIdx.x;
  const u64 mine = t < 256 ? bins[255 - t] : 0ull;
  u64 tot;
  const u64 in_front = base + block_exclusive_scan<u64>(mine, scratch, &tot);
  if (t < 256 && mine > 0 && in_front < target && in_front + mine >= target) {
    *out_bin = 255 - t;
    *out_above = in_front;
  }
  __syncthreads();
}

__global__ __launch_bounds__(NT) void td_sample_top_p_kernel(const bf16_t* __restrict__ logits, long long ld, int vocab,
                                                             float temperature, float top_p, u64 seed, u64 offset, int* __restrict__ out) {
  __shared__ u64 hist[NW][256];          // per-wave mass histograms (32 KiB)
  __shared__ unsigned cnt[NW][256];      // per-wave count histograms (16 KiB), pass 2 only
  __shared__ u64 bins[256];
  __shared__ unsigned cbins[256];
  __shared__ u64 scr64[NW + 1];
  __shared__ unsigned scr32[NW + 1];
  __shared__ int s_bin;
  __shared__ u64 s_above;
  __shared__ unsigned s_red[NW];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int row = blockIdx.x;
  const u32x4_t* src = (const u32x4_t*)(logits + (size_t)row * ld);
  const int nchunk = vocab >> 3;                     // 8 logits per 16-byte chunk

  // ---- pass 0: row maximum (as a key) ------------------------------------------------------------------------------
  unsigned kmax = 0;
  for (int c = tid; c < nchunk; c += NT) {
    const u32x4_t v = src[c];
#pragma unroll
    for (int j = 0; j < 4; ++j) kmax = max(kmax, max(key16(v[j] & 0xffffu), key16(v[j] >> 16)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o, 64));
  if (lane == 0) s_red[w] = kmax;
  __syncthreads();
  kmax = s_red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) kmax = max(kmax, s_red[i]);
  __syncthreads();

  // A +inf logit has all the mass (every finite logit weighs exp(-inf) = 0 and inf - inf is NaN): that IS the greedy answer,
  // so such a row takes the greedy path instead of the mass histograms (which would find no crossing bin).
  if (!(temperature > 0.f) || kmax == 0xff80u) {
    // greedy: the first index holding the maximum
    unsigned best = 0xffffffffu;
    for (int c = tid; c < nchunk; c += NT) {
      const u32x4_t v = src[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (key16(v[j] & 0xffffu) == kmax) best = min(best, (unsigned)(8 * c + 2 * j));
        if (key16(v[j] >> 16) == kmax) best = min(best, (unsigned)(8 * c + 2 * j + 1));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o, 64));
    if (lane == 0) s_red[w] = best;
    __syncthreads();
    if (tid == 0) {
      unsigned b = s_red[0];
      for (int i = 1; i < NW; ++i) b = min(b, s_red[i]);
      out[row] = (int)b;
    }
    return;
  }

  const float xmax = unkey16(kmax);
  const float c2 = 1.4426950408889634f / temperature;
  if (!(xmax > -INFINITY)) {                            // no finite logit in the row: nothing to weigh
    if (tid == 0) out[row] = 0;
    return;
  }

  // ---- pass 1: mass per high key byte -----------------------------------------------------------------------------------
  for (int i = tid; i < NW * 256; i += NT) (&hist[0][0])[i] = 0ull;
  __syncthreads();
  for (int c = tid; c < nchunk; c += NT) {
    const u32x4_t v = src[c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned k0 = key16(v[j] & 0xffffu), k1 = key16(v[j] >> 16);
      const u64 m0 = mass_of(k0, xmax, c2), m1 = mass_of(k1, xmax, c2);
      if (m0) atomicAdd(&hist[w][k0 >> 8], m0);
      if (m1) atomicAdd(&hist[w][k1 >> 8], m1);
    }
  }
  __syncthreads();
  if (tid < 256) {
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += hist[i][tid];
    bins[tid] = s;
  }
  __syncthreads();
  u64 Z;
  {
    u64 dummy = tid < 256 ? bins[tid] : 0ull;
    block_exclusive_scan<u64>(dummy, scr64, &Z);
  }
  u64 target = top_p >= 1.0f ? Z : (u64)((double)top_p * (double)Z);
  target = target < 1ull ? 1ull : (target > Z ? Z : target);
  if (tid == 0) { s_bin = 255; s_above = 0ull; }      // defined whatever the masses are (find_crossing writes only where it finds the bin)
  __syncthreads();
  find_crossing(bins, 0ull, target, scr64, &s_bin, &s_above);
  const int b_hi = s_bin;
  const u64 above_hi = s_above;
  __syncthreads();

  // ---- pass 2: mass and count per low key byte inside bin b_hi ----------------------------------------------------------------
  for (int i = tid; i < NW * 256; i += NT) { (&hist[0][0])[i] = 0ull; (&cnt[0][0])[i] = 0u; }
  __syncthreads();
  for (int c = tid; c < nchunk; c += NT) {
    const u32x4_t v = src[c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned k0 = key16(v[j] & 0xffffu), k1 = key16(v[j] >> 16);
      if ((int)(k0 >> 8) == b_hi) { atomicAdd(&hist[w][k0 & 255], mass_of(k0, xmax, c2)); atomicAdd(&cnt[w][k0 & 255], 1u); }
      if ((int)(k1 >> 8) == b_hi) { atomicAdd(&hist[w][k1 & 255], mass_of(k1, xmax, c2)); atomicAdd(&cnt[w][k1 & 255], 1u); }
    }
  }
  __syncthreads();
  if (tid < 256) {
    u64 s = 0;
    unsigned n = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) { s += hist[i][tid]; n += cnt[i][tid]; }
    bins[tid] = s;
    cbins[tid] = n;
  }
  __syncthreads();
  if (tid == 0) { s_bin = 255; s_above = above_hi; }
  __syncthreads();
  find_crossing(bins, above_hi, target, scr64, &s_bin, &s_above);
  const unsigned tau = ((unsigned)b_hi << 8) | (unsigned)s_bin;
  const u64 above = s_above;
  u64 m_tau = mass_of(tau, xmax, c2);                     // > 0 when the crossing bin holds mass
  m_tau = m_tau ? m_tau : 1ull;
  const unsigned n_tau = cbins[s_bin];
  u64 k_keep = (target - above + m_tau - 1) / m_tau;       // ties kept: the j-th is kept iff above + j * m_tau < target
  k_keep = k_keep < 1 ? 1 : (k_keep > n_tau ? n_tau : k_keep);
  const u64 kept = above + k_keep * m_tau;
  __syncthreads();

  // ---- pass 3: draw r in [0, kept) and locate it (token order: thread-major, any fixed order gives the same law) ---------
  unsigned my_ties = 0;
  u64 my_gt = 0;
  for (int c = tid; c < nchunk; c += NT) {
    const u32x4_t v = src[c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned k0 = key16(v[j] & 0xffffu), k1 = key16(v[j] >> 16);
      if (k0 > tau) my_gt += mass_of(k0, xmax, c2); else if (k0 == tau) ++my_ties;
      if (k1 > tau) my_gt += mass_of(k1, xmax, c2); else if (k1 == tau) ++my_ties;
    }
  }
  unsigned tie_total;
  const unsigned tie_off = block_exclusive_scan<unsigned>(my_ties, scr32, &tie_total);
  const unsigned tie_room = tie_off >= k_keep ? 0u : (unsigned)min((u64)my_ties, k_keep - tie_off);
  const u64 my_mass = my_gt + (u64)tie_room * m_tau;
  u64 mass_total;
  const u64 m_off = block_exclusive_scan<u64>(my_mass, scr64, &mass_total);
  const u64 rnd = splitmix64(splitmix64(seed ^ (0xD1B54A32D192ED03ull * (offset + 1ull))) ^ (0x9E3779B97F4A7C15ull * ((u64)row + 1ull)));
  const u64 r = __umul64hi(rnd, kept);                       // uniform in [0, kept); mass_total == kept by construction
  if (tid == 0) out[row] = 0;                              // always defined; the owner of r overwrites it below
  __syncthreads();
  if (my_mass > 0 && r >= m_off && r < m_off + my_mass) {
    u64 run = m_off;
    unsigned ties_seen = 0;
    int pick = -1;
    for (int c = tid; c < nchunk && pick < 0; c += NT) {
      const u32x4_t v = src[c];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned k = key16((e & 1) ? (v[e >> 1] >> 16) : (v[e >> 1] & 0xffffu));
        u64 m = 0;
        if (k > tau) m = mass_of(k, xmax, c2);
        else if (k == tau) { if (ties_seen < tie_room) m = m_tau; ++ties_seen; }
        if (pick < 0 && m > 0 && r < run + m) pick = 8 * c + e;
        run += m;
      }
    }
    if (pick >= 0) out[row] = pick;
  }
}

}  // namespace

int td_sample_top_p_launch(const bf16_t* logits, long long ld, int rows, int vocab, float temperature, float top_p,
                           unsigned long long seed, unsigned long long offset, int* out_ids, hipStream_t stream) {
  TD_CHECK_ARG(rows > 0 && rows <= 65535 && vocab >= 8, "td_sample_top_p: need 1..65535 rows and vocab >= 8 (got %d, %d)", rows, vocab);
  TD_CHECK_ARG(vocab % 8 == 0 && ld % 8 == 0 && ((uintptr_t)logits) % 16 == 0, "td_sample_top_p: vocab and row stride must be multiples of 8, logits 16-byte aligned");
  TD_CHECK_ARG(top_p > 0.f, "td_sample_top_p: top_p must be positive (got %g)", (double)top_p);
  hipLaunchKernelGGL(td_sample_top_p_kernel, dim3(rows), dim3(NT), 0, stream, logits, ld, vocab, temperature, top_p, seed, offset, out_ids);
  TD_CHECK_LAUNCH();
  return 0;
}
