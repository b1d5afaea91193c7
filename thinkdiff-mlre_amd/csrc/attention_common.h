// Shared pieces of the gfx950 attention kernels (attention_bf16.hip: the 8-wave forms; attention_pp.hip: the 4-wave form with two
// row blocks per wave): tile constants, lane-exchange helpers, the tile math (reference point and row sums on the matrix pipe),
// the 16-byte epilogue store and the layout of the stream-K hand-off workspace.  Internal; included by those two files only.
#pragma once
#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

#include "td_common.h"
#include "td_kernels.h"

namespace {

constexpr int D = 128;          // head dim
constexpr int KV_TILE = 64;     // keys per iteration
constexpr int Q_WAVE = 32;      // query rows per wave
constexpr int TILE_BYTES = KV_TILE * D * 2;  // 16 KiB per K or V tile

// v_permlane32_swap a, b: lanes 32-63 of a <-> lanes 0-31 of b.  Starting from a == b == x this
// leaves a = {x.lo, x.lo}, b = {x.hi, x.hi}: every lane then sees both halves of its row.
// (Inline asm on two distinct registers: hipcc folds the builtin called with identical operands.)
__device__ __forceinline__ void half_swap(float x, float& lo, float& hi) {
  lo = x;
  hi = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
}
// v_max3_f32 as ONE instruction: fmaxf on MFMA outputs makes hipcc emit a canonicalising v_max per operand
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float half_swap_max(float x) {
  float a, b;
  half_swap(x, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float half_swap_sum(float x) {
  float a, b;
  half_swap(x, a, b);
  return a + b;
}

// The smallest bf16-representable value >= x: a positive magnitude rounds up, a negative one is truncated.
__device__ __forceinline__ float bf16_ceil(float x) {
  const unsigned u = as_u32(x);
  return as_f32((u & 0x80000000u) ? (u & 0xffff0000u) : ((u + 0xffffu) & 0xffff0000u));
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>) (inline-asm immediates need constants)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void attn_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    attn_static_for<N, I + 1>(f);
  }
}
// ds_read_b64_tr_b16 as inline asm (see attn_tile_softmax_pv); the caller waits with a counted lgkmcnt before the first use
template <unsigned OFF>
__device__ __forceinline__ bf16x4_t attn_ds_read_tr16(const unsigned addr) {
  bf16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// ---- the tile body shared by the two kernels -------------------------------------------------------------------------------
// Round 3: the softmax's per-score VALU work that is NOT the exponential moved onto the matrix pipe (rocprofv3 SQ counters of
// round 2: 6.0 VALU per MFMA, the SIMD's issue slots -- not the half-busy matrix pipe -- set the pace):
//  * the running reference m of a query row enters the scores INSIDE the QK^T accumulation: one extra k-step whose K-side
//    fragment is the constant column 1 and whose Q-side fragment holds -m (m is kept bf16-representable, so the product is exact;
//    softmax is invariant to the reference point, any m serves) -- the accumulator leaves the MFMA chain as s - m and the
//    per-score `fma(s, c, -m c)` disappears (32 per tile and lane), with q pre-multiplied by scale * log2(e) where its producer
//    rounds it to bf16 anyway (TdAttnParams::q_prescaled; otherwise one multiply per score remains);
//  * the row sums come from the P.V product itself: one more output row-block whose V^T rows are all ones (4 MFMAs per tile
//    into a 16-register accumulator, every register the complete sum over the 64 keys) replaces 32 adds per tile and lane and
//    the half swap at the end; the sum is over the bf16-rounded probabilities the P.V product consumes.
// Per 64-key tile and wave: 38 MFMAs (was 32) against ~32 v_exp + 16 v_cvt_pk + 16 v_max3 (+ rare rescales).
// NOREF: no reference k-step (the FIXED form: the caller's score bound keeps exp2(s) itself inside the floating-point range, so the scores
// are exponentiated as they are) -- 36 MFMAs per tile instead of 38.
template <unsigned PO, bool NOREF = false>
__device__ __forceinline__ void attn_tile_scores(f32x16_t (&st)[2], const bf16x8_t (&qf)[8], const bf16x8_t kone, const bf16x8_t qnegm,
                                                 const unsigned (&ka)[8]) {
  // K fragments are fetched KPF MFMAs ahead of their use (pinned: hipcc would issue each read right before its consumer and
  // expose the LDS latency 16 times per tile); depth re-measured in-process at S = 4289: 2 beats 4 and 6 by 2 % (1 ties)
  constexpr int KPF = 2;
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto kread = [&](int e) {   // e = kb * 8 + ks
    return *(const TD_LDS bf16x8_t*)(uintptr_t)(ka[e & 7] + PO + (e >> 3) * 32 * 256);
  };
  bf16x8_t kf[16];
#pragma unroll
  for (int e = 0; e < KPF; ++e) kf[e] = kread(e);
  __builtin_amdgcn_sched_group_barrier(0x100, KPF, 0);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    if (!NOREF && (e & 7) == 0) {      // the reference point first: st = 1 . (-m)
      st[e >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kone, qnegm, zero16, 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    if (e + KPF < 16) kf[e + KPF] = kread(e + KPF);
    st[e >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[e], qf[e & 7], (NOREF && (e & 7) == 0) ? zero16 : st[e >> 3], 0, 0, 0);
    if (e + KPF < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  }
}

// st holds s - m_run (raw score units; log2 units when PRE).  `first`: the first tile of this (part of an) item -- o, lacc are
// zero and m_run is 0: the reference point is set to the tile's row maximum whatever it is.
// RS: row sums on the matrix pipe (lacc); otherwise they are added on the VALU into lacc[0] as HALF sums (the caller adds the
// two lane halves at the end) -- the A/B of which pipe has room on a given shape.
// FIXED: the caller knows a bound of every score (TdAttnParams::score_bound <= 48 octaves): the scores are exponentiated as they are -- no row
// maximum, no branch, no rescale, no reference at all (m_run = 0).  A bf16 probability keeps its 8 mantissa bits at any magnitude and the sums
// are fp32: exp2(s) <= 2^48, a row sum <= 2^61, and exp2(s) >= 2^-48 (|s| <= bound) -- all far inside the floating-point range.
template <unsigned PO, bool PRE, bool RS = true, bool FIXED = false>
__device__ __forceinline__ void attn_tile_softmax_pv(f32x16_t (&st)[2], f32x16_t (&o)[4], f32x16_t& lacc, float& m_run, bf16x8_t& qnegm,
                                                     const bool first, const float c, const unsigned (&va)[2][4], const int h5) {
  const float cc = PRE ? 1.0f : c;
  if constexpr (!FIXED) {
  // The FIRST read of the fresh QK^T accumulators is one the compiler can see (fmaxf): its hazard recogniser then places the wait states a
  // VALU read of a 16-pass MFMA result needs.  Inline-asm v_max3 alone is invisible to it and may read registers the matrix pipe has
  // not written yet (csrc/attention_fp8.hip met exactly that: a row maximum that missed elements, about one row in 600).
  float mx = fmaxf(st[0][15], st[1][15]);
#pragma unroll
  for (int r = 0; r < 15; ++r) mx = max3(mx, st[0][r], st[1][r]);
  mx = half_swap_max(mx);
  constexpr float RESCALE_LOG2 = 8.0f;        // deferred rescale: probabilities may reach 2^8 before the reference moves
  if (first || __any(mx * cc > RESCALE_LOG2)) {
    float target = first ? mx : fmaxf(mx, 0.f);             // new reference relative to the old one
    if (!(target > -INFINITY)) target = 0.f;                  // a fully masked row keeps its reference
    const float m_new = bf16_ceil(m_run + target);
    const float d = m_new - m_run;                            // exact: both are short bf16 values
    if (!first) {
      const float alpha = __builtin_amdgcn_exp2f(-d * cc);
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      if constexpr (RS) {
#pragma unroll
        for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
      } else lacc[0] *= alpha;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] -= d;
    m_run = m_new;
    qnegm[0] = h5 == 0 ? (short)f2bf(-m_new) : (short)0;
  }
  }      // (!FIXED)
  bf16x8_t pf[2][2];
  float psum = 0.f;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4_t pk;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = st[kb][8 * s + 2 * j], x1 = st[kb][8 * s + 2 * j + 1];
        const float p0 = __builtin_amdgcn_exp2f(PRE ? x0 : x0 * c);
        const float p1 = __builtin_amdgcn_exp2f(PRE ? x1 : x1 * c);
        if constexpr (!RS) psum += p0 + p1;
        pk[j] = pack_bf2(p0, p1);
      }
      pf[kb][s] = __builtin_bit_cast(bf16x8_t, pk);
    }
  }
  if constexpr (!RS) lacc[0] += psum;

  // ---- O^T += V^T . P^T, and the row sums as one more row-block of ones ---------------------------------------------------
  // The transposed V^T reads are INLINE ASM, with their own counted waits: behind an LDS-DMA in flight hipcc puts `s_waitcnt vmcnt(0)`
  // in front of the first __builtin_amdgcn_ds_read_tr16_b64 (its memory operand may alias the DMA's destination as far as the
  // compiler can tell; plain loads through an integer-cast address carry no such operand and are left alone), i.e. every wave waited
  // for the NEXT tile's K | V in the middle of the running tile -- and the two-slot kernels had come to RELY on that wait (see their tile top).  LDS operations of a wave complete in order,
  // so `lgkmcnt(n)` with n = the reads issued behind the wanted pair is exact, and a compiler-inserted wait that does not know of
  // these reads can only wait longer than it means to.  The wait carries the fragment as an in/out operand so that its MFMA cannot
  // be scheduled above it; sched_barrier pins the read / wait / MFMA order (VALU and SALU work may still move across).
  constexpr int VPF = 2;   // V^T fragments in flight ahead of their MFMA (2 transposed reads each)
  const bf16x8_t ones8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  bf16x4_t v0[16], v1[16];
  attn_static_for<VPF>([&](auto ec) {
    constexpr int e = decltype(ec)::value;
    v0[e] = attn_ds_read_tr16<PO + (e >> 2) * 16 * 256>(va[0][e & 3]);
    v1[e] = attn_ds_read_tr16<PO + (e >> 2) * 16 * 256>(va[1][e & 3]);
  });
  attn_static_for<16>([&](auto ec) {
    constexpr int e = decltype(ec)::value;
    if constexpr (e + VPF < 16) {
      v0[e + VPF] = attn_ds_read_tr16<PO + ((e + VPF) >> 2) * 16 * 256>(va[0][(e + VPF) & 3]);
      v1[e + VPF] = attn_ds_read_tr16<PO + ((e + VPF) >> 2) * 16 * 256>(va[1][(e + VPF) & 3]);
    }
    constexpr int behind = 2 * (15 - e < VPF ? 15 - e : VPF);      // transposed reads issued behind pair e
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(v0[e]), "+v"(v1[e]) : "n"(behind));
    bf16x8_t vf;
    vf[0] = v0[e][0]; vf[1] = v0[e][1]; vf[2] = v0[e][2]; vf[3] = v0[e][3];
    vf[4] = v1[e][0]; vf[5] = v1[e][1]; vf[6] = v1[e][2]; vf[7] = v1[e][3];
    o[e & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[e >> 3][(e >> 2) & 1], o[e & 3], 0, 0, 0);
    if constexpr (RS && (e & 3) == 3) lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pf[e >> 3][(e >> 2) & 1], lacc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0x006);
  });
}

// Normalise and store one wave's 32 x 128 output tile.  A lane holds O[q][db*32 + 8 g + 4 h5 + (0..3)] for g = 0..3: with the two
// lane halves of a row exchanged pairwise (v_permlane32_swap: upper half of group g <-> lower half of group g+1) every lane owns 8
// contiguous columns = one 16-byte store, 8 per lane instead of 16 of 8 bytes: the store tail of an attention workgroup is bound
// by the number of store instructions, not by bytes (guide, T21).
// `live`: the lane's row exists (rows past Sq take part in the lane exchange -- the swap needs both halves -- and skip the store).
__device__ __forceinline__ void attn_store_rows(const f32x16_t (&o)[4], const float inv, bf16_t* row_ptr, const int h5, const bool wide, const bool live) {
  if (wide) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        unsigned a0 = pack_bf2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv), a1 = pack_bf2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        unsigned b0 = pack_bf2(o[db][4 * g + 4] * inv, o[db][4 * g + 5] * inv), b1 = pack_bf2(o[db][4 * g + 6] * inv, o[db][4 * g + 7] * inv);
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        const u32x4_t w = {r0[0], r1[0], r0[1], r1[1]};
        if (live) *(u32x4_t*)(row_ptr + db * 32 + 8 * (g + h5)) = w;
      }
  } else {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t w;
        w[0] = pack_bf2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
        w[1] = pack_bf2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        if (live) *(u32x2_t*)(row_ptr + 4 * h5 + db * 32 + 8 * g) = w;
      }
  }
}

// The same tile as symmetric int8 under a per-row scale fixed in advance (the FLUX engine's history-scaled int8 mode: the attention output
// is the A operand of the next int8 GEMM): q = clamp(rint(bf16(o * inv) * qinv), +-127), 8 contiguous bytes per lane after the same pairwise
// lane exchange, and the row-head maximum of |bf16(o * inv)| joins amax (atomic max on the float bits; lanes of the lower half only).
__device__ __forceinline__ void attn_store_rows_q8(const f32x16_t (&o)[4], const float inv, uint8_t* row_ptr, const int h5, const float qinv, unsigned* amax,
                                                   const bool live) {
  float am = 0.f;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      unsigned w[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned acc = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float v = rbf(o[db][4 * (g + h) + b] * inv);          // the attention output is a bf16 tensor in the reference graph
          am = fmaxf(am, fabsf(v));
          acc |= ((unsigned)__float2int_rn(fminf(fmaxf(v * qinv, -127.f), 127.f)) & 0xffu) << (8 * b);
        }
        w[h] = acc;
      }
      const auto r = __builtin_amdgcn_permlane32_swap(w[0], w[1], false, false);
      if (live) *(u32x2_t*)(row_ptr + db * 32 + 8 * (g + h5)) = u32x2_t{r[0], r[1]};
    }
  am = half_swap_max(am);
  if (live && h5 == 0) __hip_atomic_fetch_max(amax, as_u32(am), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

int td_attn_set_lds_attr(const void* kernel, int bytes, std::atomic<unsigned long long>& done, int dev);

namespace {
constexpr int SK_SLOT_FLOATS = 8 * 64 * 68;          // per boundary and side: 8 waves x 64 lanes x (64 O + m + l + 2 pad) floats
constexpr int SK_HEADER_BYTES = 4096;                 // cnt[j] at word 16 + j
constexpr int SK_MAX_RANGES = SK_HEADER_BYTES / 4 - 16;
// workspace = [header | T slots: ranges x SK_SLOT_FLOATS | H slots: ranges x SK_SLOT_FLOATS]
}  // namespace
