// KV-cached decode attention (one query token per sequence) for gfx950.
//
// The prefill kernel (attention_bf16.hip) tiles 256 query rows per workgroup; with Sq = 1 it wastes 255 of them and walks the
// cache one 64-key tile at a time in a single workgroup per head (13 us for 300 keys).  Here a workgroup owns one
// (sequence, q head): its 4 waves split the keys 16 at a time -- a key row is read once by 16 lanes (16 B each, the whole
// 256-B row coalesced) -- every 16-lane key slot keeps its own
// online-softmax state and output slice in registers, and the 16 slots are merged at the end (shuffles inside a wave, LDS
// across waves).  fp32 scores / softmax / accumulation; q.k through v_dot2c_f32_bf16.
// Semantics = td_attn_launch with Sq = 1, causal, kv_lens (keys [0, kv_lens[b]) of sequence b are visible).
#include <atomic>

#include "td_common.h"
#include "td_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ float dot8q(const u32x4_t& a, const u32x4_t& b) {
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned ua = a[q], ub = b[q];   // copy out first: bit_cast on a vector-element lvalue reads element 0 (hipcc 7.2)
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ua), __builtin_bit_cast(bf16x2_t, ub), acc, false);
  }
  return acc;
}

// Sum over the 16 lanes of a DPP row, every lane ending with the total: four rotate-and-add VALU instructions (row_ror 8, 4, 2, 1).  The
// `__shfl_xor(s, off, 16)` butterfly this replaces compiles to ds_bpermute -- four DEPENDENT trips through the LDS crossbar per score, ~230 per
// workgroup at 300 keys and three query heads: that chain, not the K/V stream, was most of the kernel's 23 us at 64 sequences (round 4).
__device__ __forceinline__ float row_sum16(float s) {
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x128, 0xf, 0xf, false));
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x124, 0xf, 0xf, false));
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x122, 0xf, 0xf, false));
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x121, 0xf, 0xf, false));
  return s;
}

template <int G>
__global__ __launch_bounds__(256) void td_attn_decode_kernel(const TdAttnParams p) {
  __shared__ float sm_m[4][G], sm_l[4][G];
  __shared__ float sm_o[4][G][128];
  // one workgroup per (q-head group of G, sequence); G = 1 launches one per q head: K/V of a kv head are then read by each of
  // its q heads (from L2), which is cheap next to the parallelism it buys at small batch
  const int qg = blockIdx.x, b = blockIdx.y;
  const int kvh = (qg * G) / p.q_per_kv;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 15;                       // 16-B chunk of the 256-B head row
  const int slot = wid * 4 + (lane >> 4);        // key slot 0..15
  const int len = p.kv_lens ? p.kv_lens[b] : p.Skv;
  const int cslot = p.dec_slots ? p.dec_slots[b] : b;      // the cache slot of sequence b
  const bf16_t* Kb = p.K + (size_t)cslot * p.kv_bstride + (size_t)kvh * 128 + 8 * j;
  const bf16_t* Vb = p.V + (size_t)cslot * p.kv_bstride + (size_t)kvh * 128 + 8 * j;
  const bf16_t* Qb = p.Q + (size_t)b * p.q_bstride + (size_t)qg * G * 128 + 8 * j;
  u32x4_t q[G];
#pragma unroll
  for (int g = 0; g < G; ++g) q[g] = *(const u32x4_t*)(Qb + (size_t)g * 128);
  // fused rotary embedding + cache write (TdAttnParams::dec_kv_new): lane j of a 16-lane row holds head dims 8j .. 8j+7; rotate_half pairs dim d with
  // d +- 64, i.e. lane j with lane j ^ 8 of the same row (one DPP rotate by 8)
  const bool fused = p.dec_kv_new != nullptr;
  u32x4_t knew = {0u, 0u, 0u, 0u}, vnew = {0u, 0u, 0u, 0u};
  if (fused) {
    float cs[8], sn[8];
    {
      const f32x4_t c0 = *(const f32x4_t*)(p.dec_cos + (size_t)b * 128 + 8 * j), c1 = *(const f32x4_t*)(p.dec_cos + (size_t)b * 128 + 8 * j + 4);
      const f32x4_t s0 = *(const f32x4_t*)(p.dec_sin + (size_t)b * 128 + 8 * j), s1 = *(const f32x4_t*)(p.dec_sin + (size_t)b * 128 + 8 * j + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { cs[i] = c0[i]; cs[4 + i] = c1[i]; sn[i] = s0[i]; sn[4 + i] = s1[i]; }
    }
    auto rope = [&](const u32x4_t raw) -> u32x4_t {
      float x[8], y[8];
      unsigned r4[4] = {raw[0], raw[1], raw[2], raw[3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) { x[2 * i] = bf_lo(r4[i]); x[2 * i + 1] = bf_hi(r4[i]); }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float other = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[i]), 0x128, 0xf, 0xf, false));      // lane j ^ 8 of the row
        const float rot = (j < 8) ? -other : other;
        y[i] = rbf(x[i] * cs[i]) + rbf(rot * sn[i]);
      }
      return u32x4_t{pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3]), pack_bf2(y[4], y[5]), pack_bf2(y[6], y[7])};
    };
#pragma unroll
    for (int g = 0; g < G; ++g) q[g] = rope(q[g]);
    const int KVW = p.Hkv * 2 * 128;
    knew = rope(*(const u32x4_t*)(p.dec_kv_new + (size_t)b * KVW + (size_t)kvh * 128 + 8 * j));
    vnew = *(const u32x4_t*)(p.dec_kv_new + (size_t)b * KVW + (size_t)(p.Hkv + kvh) * 128 + 8 * j);
    if ((qg * G) % p.q_per_kv == 0 && tid < 16) {      // one 16-lane row of the kv head's first workgroup writes the new cache row
      bf16_t* dst = (bf16_t*)p.K + (size_t)p.dec_row_off[b] * p.ldkv;      // (a ROW index: 256 sequences x 8192 rows x 1024 elements pass 2^31)
      *(u32x4_t*)(dst + (size_t)kvh * 128 + 8 * j) = knew;
      *(u32x4_t*)(dst + (size_t)(p.Hkv + kvh) * 128 + 8 * j) = vnew;
    }
  }
  const int len_cache = fused ? len - 1 : len;      // keys that are read from the cache; with the fused form the last key sits in registers
  float m[G], l[G], o[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[g][i] = 0.f;
  }
  const float sc = p.scale * 1.4426950408889634f;   // softmax in base 2
  // one key of this slot: scores of the G query heads against it, online softmax, value accumulation
  auto one_key = [&](const u32x4_t& kk, const u32x4_t& vv) {
    float vf[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { vf[2 * i] = bf_lo(vv[i]); vf[2 * i + 1] = bf_hi(vv[i]); }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float s = row_sum16(dot8q(kk, q[g])) * sc;
      const float mn = fmaxf(m[g], s);
      const float corr = __builtin_amdgcn_exp2f(m[g] - mn), pe = __builtin_amdgcn_exp2f(s - mn);
      m[g] = mn;
      l[g] = l[g] * corr + pe;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[g][i] = o[g][i] * corr + pe * vf[i];
    }
  };
  // The loop is a chain of dependent HBM / L2 round trips (one key row per slot and trip: ~0.6 us each, 19 trips for 300 keys = the 12 us the
  // kernel took per layer at one sequence): four keys per slot are fetched before the first is used, so a trip's latency covers four keys.
  // The keys of a slot are still visited in increasing order, so the arithmetic -- and the result -- is unchanged.
  // ... and the NEXT four are requested before the current four are used (register double buffer): with several query heads per workgroup the
  // arithmetic of a trip (~35 instructions per key and head) is as long as its fetch, and the two alternated instead of overlapping.
  constexpr int UN = G >= 6 ? 2 : 4;      // (6 / 7 heads per workgroup: two keys per buffer keep the kernel inside 256 registers, i.e. two waves per SIMD)
  int key = slot;
  if constexpr (G == 1) {      // one head per workgroup (small batches: the grid is heads x sequences): a trip is all fetch, nothing to overlap it with
    for (; key + 16 * (UN - 1) < len_cache; key += 16 * UN) {
      u32x4_t kk[UN], vv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        kk[u] = *(const u32x4_t*)(Kb + (size_t)(key + 16 * u) * p.ldkv);
        vv[u] = *(const u32x4_t*)(Vb + (size_t)(key + 16 * u) * p.ldkv);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) one_key(kk[u], vv[u]);
    }
  } else if (key + 16 * (UN - 1) < len_cache) {
    u32x4_t kk[UN], vv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      kk[u] = *(const u32x4_t*)(Kb + (size_t)(key + 16 * u) * p.ldkv);
      vv[u] = *(const u32x4_t*)(Vb + (size_t)(key + 16 * u) * p.ldkv);
    }
    for (;;) {
      const int nkey = key + 16 * UN;
      const bool more = nkey + 16 * (UN - 1) < len_cache;
      u32x4_t kn[UN], vn[UN];
      if (more) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          kn[u] = *(const u32x4_t*)(Kb + (size_t)(nkey + 16 * u) * p.ldkv);
          vn[u] = *(const u32x4_t*)(Vb + (size_t)(nkey + 16 * u) * p.ldkv);
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) one_key(kk[u], vv[u]);
      key = nkey;
      if (!more) break;
#pragma unroll
      for (int u = 0; u < UN; ++u) { kk[u] = kn[u]; vv[u] = vn[u]; }
    }
  }
  for (; key < len_cache; key += 16) {
    const u32x4_t kk = *(const u32x4_t*)(Kb + (size_t)key * p.ldkv);
    const u32x4_t vv = *(const u32x4_t*)(Vb + (size_t)key * p.ldkv);
    one_key(kk, vv);
  }
  if (fused && key == len - 1) one_key(knew, vnew);      // the slot that owns key len - 1 in the strided order: same arithmetic order as reading it from the cache
  // merge the 4 key slots of this wave (lanes 16 / 32 apart hold the same d-chunk), then the 4 waves through LDS
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
      const float mo = __shfl_xor(m[g], off, 64), lo = __shfl_xor(l[g], off, 64);
      const float mn = fmaxf(m[g], mo);
      const float ca = m[g] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m[g] - mn);
      const float cb = mo == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mo - mn);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[g][i] = o[g][i] * ca + __shfl_xor(o[g][i], off, 64) * cb;
      l[g] = l[g] * ca + lo * cb;
      m[g] = mn;
    }
    if (lane < 16) {
      if (lane == 0) { sm_m[wid][g] = m[g]; sm_l[wid][g] = l[g]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) sm_o[wid][g][8 * j + i] = o[g][i];
    }
  }
  __syncthreads();
  // 256 threads: thread t writes output elements (g, d) for t = g' * 128 + d over ceil(G*128/256) passes
  bf16_t* Ob = p.O + (size_t)b * p.o_bstride + (size_t)qg * G * 128;
  for (int e = tid; e < G * 128; e += 256) {
    const int g = e >> 7, d = e & 127;
    float mn = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mn = fmaxf(mn, sm_m[w][g]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float c = sm_m[w][g] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(sm_m[w][g] - mn);
      num += sm_o[w][g][d] * c;
      den += sm_l[w][g] * c;
    }
    Ob[e] = f2bf(den > 0.f ? num / den : 0.f);
  }
}

}  // namespace

static std::atomic<int> g_decode_group{0};
extern "C" int td_attention_decode_set_group(int g) { return g_decode_group.exchange(g); }

int td_attn_decode_launch(const TdAttnParams& p, hipStream_t stream) {
  TD_CHECK_ARG(p.Sq == 1 && p.head_dim == 128 && p.Hq % p.Hkv == 0, "td_attn_decode: needs Sq = 1, head_dim 128");
  TdAttnParams q = p;
  q.q_per_kv = p.Hq / p.Hkv;
  if (p.dec_kv_new) TD_CHECK_ARG(p.kv_lens && p.dec_cos && p.dec_sin && p.dec_row_off && ((uintptr_t)p.dec_kv_new | (uintptr_t)p.dec_cos | (uintptr_t)p.dec_sin) % 16 == 0,
                                 "td_attn_decode: the fused rotary / cache-write form needs kv_lens, both table rows and the cache row offsets");
  TD_CHECK_ARG(p.ldkv % 8 == 0 && ((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V | (uintptr_t)p.O) % 16 == 0 && p.q_bstride % 8 == 0 && p.kv_bstride % 8 == 0,
               "td_attn_decode: operands must be 16-byte aligned");
  // G q heads of one kv head per workgroup read its K/V once instead of G times (from L2); taken when the grid still gives
  // every CU a workgroup -- many sequences -- and left at one head per workgroup for small batches
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    static std::atomic<int> cached[64] = {};
    int n = cached[dev & 63].load(std::memory_order_relaxed);
    if (n == 0 && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cached[dev & 63].store(n, std::memory_order_relaxed);
    if (n > 0) cus = n;
  }
  const int forced = g_decode_group.load(std::memory_order_relaxed);      // td_attention_decode_set_group: tests and A/B
  int G = 1;
  for (int g : {7, 6, 4, 3, 2})
    if (q.q_per_kv % g == 0 && (forced ? g == forced : (long long)(p.Hq / g) * p.batch >= cus)) { G = g; break; }
  const dim3 grid(p.Hq / G, p.batch);
  switch (G) {
    case 7: hipLaunchKernelGGL(td_attn_decode_kernel<7>, grid, dim3(256), 0, stream, q); break;
    case 6: hipLaunchKernelGGL(td_attn_decode_kernel<6>, grid, dim3(256), 0, stream, q); break;
    case 4: hipLaunchKernelGGL(td_attn_decode_kernel<4>, grid, dim3(256), 0, stream, q); break;
    case 3: hipLaunchKernelGGL(td_attn_decode_kernel<3>, grid, dim3(256), 0, stream, q); break;
    case 2: hipLaunchKernelGGL(td_attn_decode_kernel<2>, grid, dim3(256), 0, stream, q); break;
    default: hipLaunchKernelGGL(td_attn_decode_kernel<1>, grid, dim3(256), 0, stream, q);
  }
  TD_CHECK_LAUNCH();
  return 0;
}
