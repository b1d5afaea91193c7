// Flash-style fused attention forward for gfx950, head_dim = 128, bf16 in/out, fp32 softmax state.
//
// Serves K11: FLUX joint attention over [text || image] tokens (non-causal, no
// mask, 24 heads, S ~ 4.3k) and K19: Qwen2-VL prefill attention (causal, GQA 28q/4kv).
// Replaces F.scaled_dot_product_attention inside [ext] diffusers FluxAttnProcessor2_0 /
// FluxSingleAttnProcessor2_0 and the vLLM fork's Qwen2-VL attention.
//
// Structure (cdna_hip_programming guide, Appendix B "Fused attention prefill"):
//  * workgroup = 8 waves, each wave owns 32 query rows (256 rows / workgroup); KV tile = 64 keys.
//  * swapped QK^T: S^T[key][q] = K . Q^T on v_mfma_f32_32x32x16_bf16, so one query row lives on
//    one lane (and its lane+32 partner): the online softmax is in-register, the row max / row sum
//    need one half-swap (v_permlane32_swap) and no LDS.
//  * the S^T accumulator, converted to bf16 in place, IS the B operand of the P.V product
//    (O^T[d][q] += V^T[d][key] . P^T[key][q]); V^T fragments come from a row-major V tile through
//    ds_read_b64_tr_b16 (hardware transpose read).
//  * K/V tiles travel HBM -> LDS by LDS-DMA (buffer_load ... lds), double buffered; both tiles use
//    the one 256-B-row XOR image that is conflict-free for row reads and transposed reads
//    (off(row,ch) = 256 row + 16 (ch ^ ((row&3)<<2 | (row>>2)&3))), applied on the DMA source side.
//  * q/k/v are read in place from the projection output ([S, ld] rows, head h at column h*128):
//    no head-major re-layout pass exists anywhere on the path.
#include "attention_common.h"

// ---------------------------------------------------------------------------------------------
// One workgroup per (256-row query tile, head, batch).  Lean instruction stream: rocprofv3 PMC on the first form of this
// kernel (S=4289, 24 heads) showed 7.2 VALU instructions per MFMA and the SIMD's issue slots, not the matrix pipe (50 % busy),
// setting the pace -- two thirds of that VALU work was LDS address arithmetic (XOR-swizzled addresses recomputed per read) and
// accumulator zeroing.  Here every per-lane LDS address lives in a register for the whole kernel (8 for the K row reads,
// 8 for the V transposed reads; k-block / k-step / tile-slot parts are instruction immediates).
// ---------------------------------------------------------------------------------------------
template <bool CAUSAL, int NWAVES, bool BIAS = false, bool VARLEN = false, bool PRE = false, bool RS = true, bool FIXED = false>
__global__ __launch_bounds__(NWAVES * 64, 2) void td_attn_fwd_d128_lean_kernel(const TdAttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [K slot 0 | K slot 1 | V slot 0 | V slot 1]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5;
  const int l31 = lane & 31;

  const int qblk = blockIdx.x;
  const int head = blockIdx.y;
  const int batch = blockIdx.z;
  const int kvhead = head / p.q_per_kv;
  const int q0 = qblk * (NWAVES * Q_WAVE) + wid * Q_WAVE;

  const bf16_t* Qb = p.Q + (size_t)batch * p.q_bstride;
  const bf16_t* Kb = p.K + (size_t)batch * p.kv_bstride;
  const bf16_t* Vb = p.V + (size_t)batch * p.kv_bstride;
  bf16_t* Ob = p.O + (size_t)batch * p.o_bstride;
  int Sq = p.Sq, Skv = p.Skv, c_off = p.causal_offset;
  if constexpr (VARLEN) {
    // packed segments (the vision towers' cu_seqlens): segment `batch` = rows [seg_starts[b], seg_starts[b+1]) of q, k, v and o,
    // full attention inside it; the grid is sized for the longest segment, workgroups past this one's rows leave at once
    const int r0 = p.seg_starts[batch];
    Sq = Skv = p.seg_starts[batch + 1] - r0;
    if (qblk * (NWAVES * Q_WAVE) >= Sq) return;
    Qb += (size_t)r0 * p.ldq; Kb += (size_t)r0 * p.ldkv; Vb += (size_t)r0 * p.ldkv; Ob += (size_t)r0 * p.ldo;
  }

  const unsigned q_bytes = (unsigned)(((long long)(Sq - 1) * p.ldq + p.Hq * D) * 2);
  // batched KV-cached decode: sequence `batch` has its own cache length (causal instantiation only, so the joint-attention
  // instruction stream of FLUX is untouched); keys visible to query row q: key <= q + (Skv - Sq)
  if constexpr (CAUSAL) {
    if (p.kv_lens) { Skv = p.kv_lens[batch]; c_off = Skv - p.Sq; }
  }
  const unsigned kv_bytes = (unsigned)(((long long)(Skv - 1) * p.ldkv + p.Hkv * D) * 2);
  __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qb, 0, q_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, kv_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, kv_bytes, 0x00020000);

  int nt = (Skv + KV_TILE - 1) / KV_TILE;
  if (CAUSAL) {
    const int last_q = min(Sq, (qblk + 1) * NWAVES * Q_WAVE) - 1 + c_off;
    nt = min(nt, last_q / KV_TILE + 1);
  }

  constexpr int GROUPS = KV_TILE / 4;
  constexpr int SG = (GROUPS + NWAVES - 1) / NWAVES;
  const int srow = lane >> 4;
  unsigned voffK[SG];
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    const int g = wid + NWAVES * s;
    const int swz = (srow << 2) | (g & 3);
    const int chunk = (lane & 15) ^ swz;
    voffK[s] = (unsigned)(g * 4 + srow) * (unsigned)p.ldkv * 2u + (unsigned)(kvhead * D + chunk * 8) * 2u;
  }
  auto stage = [&](int slot, int t) {
    const unsigned tile_off = (unsigned)t * KV_TILE * (unsigned)p.ldkv * 2u;
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const int g = wid + NWAVES * s;
      if (GROUPS % NWAVES == 0 || g < GROUPS) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (TD_LDS void*)(smem + slot * TILE_BYTES + g * 1024), 16, voffK[s] + tile_off, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (TD_LDS void*)(smem + (2 + slot) * TILE_BYTES + g * 1024), 16, voffK[s] + tile_off, 0, 0, 0);
      }
    }
  };
  stage(0, 0);

  bf16x8_t qf[8];
  {
    const unsigned qoff = (unsigned)(q0 + l31) * (unsigned)p.ldq * 2u + (unsigned)(head * D + 8 * h5) * 2u;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsQ, qoff + ks * 32, 0, 0);
      qf[ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }

  // ---- resident per-lane LDS byte addresses of slot 0 (the other slot is an immediate offset in the tile body) ----
  const unsigned lds0 = (unsigned)(uintptr_t)(TD_LDS char*)smem;   // LDS byte address of the tile area
  unsigned ka[8];      // K row read: row l31 (+32 kb as an immediate), chunk (2 ks + h5) ^ swz(row)
  {
    const unsigned ksw = ((lane & 3) << 2) | ((lane >> 2) & 3);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) ka[ks] = lds0 + l31 * 256 + (((2 * ks + h5) ^ ksw) << 4);
  }
  unsigned va[2][4];   // V transposed read [jj][db]: row 4 h5 + vq + 8 jj (+32 kb + 16 s as immediates)
  {
    const unsigned vq = (lane & 15) >> 2, vp = lane & 3;
    const unsigned vchunk = 2 * ((lane >> 4) & 1) + (vp >> 1);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const unsigned swz = (vq << 2) | ((2 * jj + h5) & 3);   // ((row&3)<<2) | ((row>>2)&3), independent of kb, s
#pragma unroll
      for (int db = 0; db < 4; ++db)
        va[jj][db] = lds0 + 2 * TILE_BYTES + (4 * h5 + vq + 8 * jj) * 256 + (((4 * db + vchunk) ^ swz) << 4) + 8 * (vp & 1);
    }
  }

  f32x16_t o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  if ((p.variant & 0x100) && wid >= 4) __builtin_amdgcn_s_setprio(1);   // static priority for the younger half (A/B switch)
  f32x16_t lacc;                      // row sums: every register holds the whole sum of its lane's query row
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
  float m_run = 0.f;                  // reference point of the row's exponentials (bf16-representable; set by the first tile)
  const bf16x8_t kone = {(short)(h5 == 0 ? 0x3F80 : 0), 0, 0, 0, 0, 0, 0, 0};   // K-side fragment of the reference k-step: column 0 = 1
  bf16x8_t qnegm = {0, 0, 0, 0, 0, 0, 0, 0};                                       // Q-side: row 0 = -m_run

  const float c = p.scale * 1.4426950408889634f;
  const int q_pos = q0 + l31 + c_off;

  // The tile body is instantiated for both double-buffer slots: the slot offset of every LDS read is then an instruction
  // immediate instead of 16 address flips (v_xor) per tile and wave.
  auto tile = [&](const int t, auto slot_tag) {
    constexpr unsigned PO = decltype(slot_tag)::value * TILE_BYTES;
    // tile t landed (MY pieces: the explicit vmcnt -- a workgroup-scope fence orders LDS traffic with lgkmcnt only, and LDS-DMA
    // completes on vmcnt; the compiler used to wait by accident, in front of its transposed-read builtin) and, behind the barrier,
    // everyone's; slot (t+1)&1 no longer read
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nt) stage((t + 1) & 1, t + 1);

    f32x16_t st[2];
    attn_tile_scores<PO, FIXED>(st, qf, kone, qnegm, ka);      // S^T - m = K . Q^T - 1 . m

    const int key0 = t * KV_TILE;
    if constexpr (BIAS) {   // additive score bias in the scaled domain: (s + bias/scale) * scale = s*scale + bias
      const float inv_scale = 1.0f / p.scale;
      const int qrow = min(q0 + l31, p.Sq - 1);
      const float* bp = p.bias + ((size_t)head * p.Sq + qrow) * p.Skv + key0 + 4 * h5;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int kbase = key0 + kb * 32 + 8 * g4 + 4 * h5;
          if (kbase + 4 <= p.Skv) {
            const f32x4_t bv = *(const f32x4_t*)(bp + kb * 32 + 8 * g4);
#pragma unroll
            for (int r = 0; r < 4; ++r) st[kb][4 * g4 + r] += bv[r] * inv_scale;
          }
        }
    }
    const bool need_mask = (key0 + KV_TILE > Skv) || (CAUSAL && key0 + KV_TILE - 1 > q0 + c_off);
    if (need_mask) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h5;
          if ((key >= Skv) || (CAUSAL && key > q_pos)) st[kb][r] = -INFINITY;
        }
    }

    attn_tile_softmax_pv<PO, PRE, RS, FIXED>(st, o, lacc, m_run, qnegm, t == 0, c, va, h5);
  };
  {
    int t = 0;
    for (; t + 1 < nt; t += 2) {
      tile(t, std::integral_constant<unsigned, 0>{});
      tile(t + 1, std::integral_constant<unsigned, 1>{});
    }
    if (t < nt) tile(t, std::integral_constant<unsigned, 0>{});
  }

  const float inv = 1.0f / (RS ? lacc[0] : half_swap_sum(lacc[0]));
  const int q = q0 + l31;
  bool stored = false;
  if constexpr (!CAUSAL && !BIAS && !VARLEN) {      // the int8 output form exists for the joint attention only (keeps it out of the other streams)
    if (p.q8) {
      const int qr = min(q, Sq - 1);
      attn_store_rows_q8(o, inv, p.q8 + (size_t)qr * p.ldq8 + head * D, h5, p.q8_inv[qr], p.q8_amax + qr, q < Sq);
      stored = true;
    }
  }
  if (!stored) attn_store_rows(o, inv, Ob + (size_t)min(q, Sq - 1) * p.ldo + head * D, h5, (p.ldo & 7) == 0 && !(p.variant & 0x200), q < Sq);
#endif
}


// ---------------------------------------------------------------------------------------------
// Stream-K form of the kernel above for joint (non-causal) attention: the launch is ONE round of persistent workgroups.
//
// FLUX at 1024^2: 17 query tiles x 24 heads = 408 workgroups of 68 KV tiles each on 256 CUs -- two rounds, the second 59 %
// empty.  Here the 408 x 68 (item, KV tile) iterations are cut into G equal contiguous ranges, one per workgroup, G = number
// of CUs.  A range starts inside an item and ends inside another, so a workgroup runs
//   [tail part of item a : KV tiles kb..nt) ] [whole items ...] [head part of item z : KV tiles 0..ke) ]
// and every item is split over at most two workgroups (a range is longer than an item).  The two owners of a split item
// combine their un-normalised online-softmax states (O accumulators, row max, row sum: 68 floats per lane) exactly as two KV
// tiles are combined inside the loop, and whoever arrives SECOND does it -- nobody ever waits:
//   * the owner of the TAIL part (it meets the item first thing in its life) writes its state to slot T[j] of the boundary j
//     between the two ranges and draws a ticket from cnt[j]; ticket 0: done, ticket 1: the other side has already left its
//     state in H[j] -- read it, combine, normalise, store;
//   * the owner of the HEAD part (it meets the item last) first looks at cnt[j]: 1 (the normal case) -- read T[j], combine with
//     the state it holds in registers, normalise, store; 0 -- leave its own state in H[j] and draw a ticket, and if that comes
//     back 1 after all, finish as above from registers.
// With no wait there is no forward-progress assumption: any number of these launches may share the chip with anything else,
// in any dispatch order (a spin version deadlocks when a consumer's producer is not resident).  The hand-off is the
// placement-independent recipe of the guide (6, Guideline 16 R1 with a counter as the flag): write-through (sc1) 16-byte
// stores, every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE agent-scope atomic add; the reader's side is
// ONE agent-scope acquire after the atomic that told it, s_waitcnt, barrier, plain loads.  The finisher zeroes cnt[j]; the
// workspace (one per engine context / stream: concurrent launches must not share it) is zeroed when it is created.
// ---------------------------------------------------------------------------------------------

template <int NWAVES, bool XCD_REMAP, bool PRE = false, bool RS = true, bool FIXED = false>
__global__ __launch_bounds__(NWAVES * 64, 2) void td_attn_fwd_d128_streamk_kernel(const TdAttnParams p, char* __restrict__ ws,
                                                                                  const int n_qblk, const int nt) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [K slot 0 | K slot 1 | V slot 0 | V slot 1 | ticket word]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5;
  const int l31 = lane & 31;
  unsigned* const cnt = (unsigned*)ws + 16;
  TD_LDS unsigned* const ticket_lds = (TD_LDS unsigned*)(smem + 4 * TILE_BYTES);

  // logical range of this workgroup: with XCD_REMAP, workgroups that share an XCD (equal blockIdx % 8 under round-robin
  // placement; speed only) take neighbouring ranges = neighbouring query tiles of the same heads = the same K/V in that L2
  const int G = gridDim.x;
  int r = blockIdx.x;
  if constexpr (XCD_REMAP) {
    const int q8 = G >> 3, r8 = G & 7, xcd = r & 7;
    r = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (r >> 3);
  }
  const long long total = (long long)n_qblk * p.Hq * p.batch * nt;
  long long it = total * r / G;
  const long long it_end = total * (r + 1) / G;

  const int Skv = p.Skv;
  constexpr int GROUPS = KV_TILE / 4;
  constexpr int SG = (GROUPS + NWAVES - 1) / NWAVES;
  const int srow = lane >> 4;
  const float c = p.scale * 1.4426950408889634f;
  const float cc = PRE ? 1.0f : c;          // scale of the stored reference points: log2 units when q arrives pre-scaled
  const bf16x8_t kone = {(short)(h5 == 0 ? 0x3F80 : 0), 0, 0, 0, 0, 0, 0, 0};   // K-side fragment of the reference k-step (attn_tile_scores)

  // ---- resident per-lane LDS byte addresses of slot 0 (the other slot is an immediate offset in the tile body) ----
  const unsigned lds0 = (unsigned)(uintptr_t)(TD_LDS char*)smem;
  unsigned ka[8];
  {
    const unsigned ksw = ((lane & 3) << 2) | ((lane >> 2) & 3);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) ka[ks] = lds0 + l31 * 256 + (((2 * ks + h5) ^ ksw) << 4);
  }
  unsigned va[2][4];
  {
    const unsigned vq = (lane & 15) >> 2, vp = lane & 3;
    const unsigned vchunk = 2 * ((lane >> 4) & 1) + (vp >> 1);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const unsigned swz = (vq << 2) | ((2 * jj + h5) & 3);
#pragma unroll
      for (int db = 0; db < 4; ++db)
        va[jj][db] = lds0 + 2 * TILE_BYTES + (4 * h5 + vq + 8 * jj) * 256 + (((4 * db + vchunk) ^ swz) << 4) + 8 * (vp & 1);
    }
  }

  if ((p.variant & 0x100) && wid >= 4) __builtin_amdgcn_s_setprio(1);   // static priority for the younger half (A/B switch)
  while (it < it_end) {
    // ---- the part of an item this workgroup runs now: KV tiles [kb, ke) of item `item` -------------------------------
    const int item = (int)(it / nt);
    const int kb = (int)(it - (long long)item * nt);
    const int ke = (int)min((long long)nt, kb + (it_end - it));
    __builtin_assume(ke > kb);       // a part is never empty
    it += ke - kb;
    const int qblk = item % n_qblk;
    const int hb = item / n_qblk;
    const int head = hb % p.Hq, batch = hb / p.Hq;
    const int kvhead = head / p.q_per_kv;
    const int q0 = qblk * (NWAVES * Q_WAVE) + wid * Q_WAVE;

    const bf16_t* Qb = p.Q + (size_t)batch * p.q_bstride;
    const bf16_t* Kb = p.K + (size_t)batch * p.kv_bstride;
    const bf16_t* Vb = p.V + (size_t)batch * p.kv_bstride;
    const unsigned q_bytes = (unsigned)(((long long)(p.Sq - 1) * p.ldq + p.Hq * D) * 2);
    const unsigned kv_bytes = (unsigned)(((long long)(Skv - 1) * p.ldkv + p.Hkv * D) * 2);
    __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qb, 0, q_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, kv_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, kv_bytes, 0x00020000);

    unsigned voffK[SG];
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const int g = wid + NWAVES * s;
      const int swz = (srow << 2) | (g & 3);
      const int chunk = (lane & 15) ^ swz;
      voffK[s] = (unsigned)(g * 4 + srow) * (unsigned)p.ldkv * 2u + (unsigned)(kvhead * D + chunk * 8) * 2u;
    }
    auto stage = [&](int slot, int t) {
      const unsigned tile_off = (unsigned)t * KV_TILE * (unsigned)p.ldkv * 2u;
#pragma unroll
      for (int s = 0; s < SG; ++s) {
        const int g = wid + NWAVES * s;
        if (GROUPS % NWAVES == 0 || g < GROUPS) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (TD_LDS void*)(smem + slot * TILE_BYTES + g * 1024), 16, voffK[s] + tile_off, 0, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (TD_LDS void*)(smem + (2 + slot) * TILE_BYTES + g * 1024), 16, voffK[s] + tile_off, 0, 0, 0);
        }
      }
    };
    __syncthreads();          // the previous part's last tile is no longer read by any wave
    stage(0, kb);

    bf16x8_t qf[8];
    {
      const unsigned qoff = (unsigned)(q0 + l31) * (unsigned)p.ldq * 2u + (unsigned)(head * D + 8 * h5) * 2u;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsQ, qoff + ks * 32, 0, 0);
        qf[ks] = __builtin_bit_cast(bf16x8_t, v);
      }
    }

    f32x16_t o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) o[db][rr] = 0.f;
    f32x16_t lacc;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) lacc[rr] = 0.f;
    float m_run = 0.f;
    bf16x8_t qnegm = {0, 0, 0, 0, 0, 0, 0, 0};
    auto tile = [&](const int t, auto slot_tag) {
      constexpr unsigned SLOT = decltype(slot_tag)::value;
      constexpr unsigned PO = SLOT * TILE_BYTES;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of tile t landed (explicit: see the plain-grid kernel) ...
      __syncthreads();                                        // ... and everyone's; the other slot is no longer read
      if (t + 1 < ke) stage(SLOT ^ 1, t + 1);

      f32x16_t st[2];
      attn_tile_scores<PO, FIXED>(st, qf, kone, qnegm, ka);

      const int key0 = t * KV_TILE;
      if (key0 + KV_TILE > Skv) {        // last tile: keys >= Skv are masked
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            const int key = key0 + kk * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h5;
            if (key >= Skv) st[kk][rr] = -INFINITY;
          }
      }

      attn_tile_softmax_pv<PO, PRE, RS, FIXED>(st, o, lacc, m_run, qnegm, t == kb, c, va, h5);
    };
    {
      int t = kb;
      for (; t + 1 < ke; t += 2) {
        tile(t, std::integral_constant<unsigned, 0>{});
        tile(t + 1, std::integral_constant<unsigned, 1>{});
      }
      if (t < ke) tile(t, std::integral_constant<unsigned, 0>{});
    }
    float l_run = RS ? lacc[0] : half_swap_sum(lacc[0]);           // the complete row sum (RS: every register of lacc holds it, in both lane halves)
    // ---- what happens to the state: leave it for the other owner, combine with the other owner's, or just finish ----------
    if (kb > 0 || ke < nt) {
      const int j = kb > 0 ? r : r + 1;                  // boundary between ranges j-1 (head part) and j (tail part)
      const bool tail = kb > 0;
      const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + SK_HEADER_BYTES), 0, (unsigned)(2 * G) * SK_SLOT_FLOATS * 4u, 0x00020000);
      const unsigned lane_off = ((unsigned)wid * 17u * 64u + (unsigned)lane) * 16u;      // [wave][chunk 0..16][lane] x 16 bytes
      const unsigned mine = (unsigned)(tail ? j : G + j) * (SK_SLOT_FLOATS * 4u) + lane_off;      // T[j] / H[j]
      const unsigned theirs = (unsigned)(tail ? G + j : j) * (SK_SLOT_FLOATS * 4u) + lane_off;
      // one lane asks, everyone hears the answer through LDS (block-uniform control flow below)
      auto ask = [&](bool draw) -> unsigned {
        __syncthreads();
        if (tid == 0)
          *ticket_lds = draw ? __hip_atomic_fetch_add(cnt + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : __hip_atomic_load(cnt + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        return *ticket_lds;
      };
      bool other_ready = !tail && ask(false) == 1u;      // head part: the tail part's owner has normally long been through
      if (!other_ready) {
        // leave the state for the other owner: write-through stores, drained by every wave, then the ticket
#pragma unroll
        for (int q4 = 0; q4 < 16; ++q4) {
          const u32x4_t v = {as_u32(o[q4 >> 2][4 * (q4 & 3)]), as_u32(o[q4 >> 2][4 * (q4 & 3) + 1]),
                             as_u32(o[q4 >> 2][4 * (q4 & 3) + 2]), as_u32(o[q4 >> 2][4 * (q4 & 3) + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, rsS, mine + q4 * 1024u, 0, 16);       // aux 16 = sc1
        }
        const u32x4_t ml = {as_u32(m_run), as_u32(l_run), 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(ml, rsS, mine + 16 * 1024u, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        other_ready = ask(true) == 1u;                   // the barrier inside orders every wave's drain before the add
        if (!other_ready) continue;                      // first to arrive: the other owner finishes the item
      }
      // second to arrive: the other state is complete and published before the atomic that told us
      if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const u32x4_t ml = __builtin_amdgcn_raw_buffer_load_b128(rsS, theirs + 16 * 1024u, 0, 0);
      const float m2 = as_f32(ml[0]), l2 = as_f32(ml[1]);
      const float mm = fmaxf(m_run, m2);
      const float a1 = __builtin_amdgcn_exp2f((m_run - mm) * cc), a2 = __builtin_amdgcn_exp2f((m2 - mm) * cc);
      // One operation order whoever merges -- fma(head, a_head, tail * a_tail) -- so the result does not depend on which owner
      // arrived second (it does under load; the sum is otherwise the same to one fp32 rounding).
      auto comb = [&](float own, float other) {
        return tail ? __builtin_fmaf(other, a2, own * a1) : __builtin_fmaf(own, a1, other * a2);
      };
      l_run = comb(l_run, l2);
#pragma unroll
      for (int q4 = 0; q4 < 16; ++q4) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsS, theirs + q4 * 1024u, 0, 0);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
          o[q4 >> 2][4 * (q4 & 3) + k4] = comb(o[q4 >> 2][4 * (q4 & 3) + k4], as_f32(v[k4]));
      }
      if (tid == 0) __hip_atomic_store(cnt + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // both tickets drawn: ready for the next launch
    }

    // ---- normalise and store: lane holds O[q][db*32 + (r&3) + 8 (r>>2) + 4 h5] ------------------------------------------
    const float inv = 1.0f / l_run;
    const int q = q0 + l31;
    if (p.q8) {
      const int qr = min(q, p.Sq - 1);
      attn_store_rows_q8(o, inv, p.q8 + (size_t)qr * p.ldq8 + head * D, h5, p.q8_inv[qr], p.q8_amax + qr, q < p.Sq);
    } else {
      attn_store_rows(o, inv, p.O + (size_t)batch * p.o_bstride + (size_t)min(q, p.Sq - 1) * p.ldo + head * D, h5, (p.ldo & 7) == 0 && !(p.variant & 0x200), q < p.Sq);
    }
  }
#endif
}

namespace {

size_t sk_ws_bytes(int ranges) { return (size_t)SK_HEADER_BYTES + (size_t)2 * ranges * SK_SLOT_FLOATS * sizeof(float); }

// Workspaces for callers that bring none (the plain C entry point): one per (device, stream), created on the first launch
// that needs it -- call td_attention_bf16 once per stream before capturing it into a hipGraph.  Concurrent launches must
// not share a workspace, and launches on one stream never overlap.
int sk_pooled_workspace(int dev, int ranges, hipStream_t stream, char** out) {
  struct Entry { int dev; hipStream_t stream; int ranges; char* ws; };
  static std::mutex mu;
  static std::vector<Entry> pool;
  std::lock_guard<std::mutex> lock(mu);
  for (auto& e : pool)
    if (e.dev == dev && e.stream == stream && e.ranges >= ranges) { *out = e.ws; return 0; }
  char* w = nullptr;
  TD_CHECK_HIP(hipMalloc((void**)&w, sk_ws_bytes(ranges)));
  // zeroed ON the launching stream: a null-stream memset is not ordered with a non-blocking stream, and the first persistent
  // launch of a new stream would race it for the hand-off counters
  TD_CHECK_HIP(hipMemsetAsync(w, 0, SK_HEADER_BYTES, stream));
  pool.push_back(Entry{dev, stream, ranges, w});
  *out = w;
  return 0;
}

int device_cus(int dev) {
  static std::atomic<int> cus[64] = {};
  int n = cus[dev & 63].load(std::memory_order_relaxed);
  if (n == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    n = prop.multiProcessorCount;
    cus[dev & 63].store(n, std::memory_order_relaxed);
  }
  return n;
}

template <typename K>
int set_lds_attr_once(K kernel, int bytes, std::atomic<unsigned long long>& done, int dev) {
  if (!((done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
    TD_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.fetch_or(1ull << (dev & 63), std::memory_order_release);
  }
  return 0;
}

}  // namespace

// shared with attention_fp8.hip
int td_attn_device_cus(int dev) { return device_cus(dev); }
int td_attn_pooled_workspace(int dev, int ranges, hipStream_t stream, char** out) { return sk_pooled_workspace(dev, ranges, stream, out); }
int td_attn_set_lds_attr(const void* kernel, int bytes, std::atomic<unsigned long long>& done, int dev) {
  if (!((done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
    TD_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.fetch_or(1ull << (dev & 63), std::memory_order_release);
  }
  return 0;
}

int td_attn_launch(const TdAttnParams& p, hipStream_t stream) {
  TD_CHECK_ARG(p.head_dim == D, "td_attention: head_dim=%d unsupported (only 128)", p.head_dim);
  TD_CHECK_ARG(p.Sq > 0 && p.Skv > 0 && p.Hq > 0 && p.Hkv > 0 && p.batch > 0, "td_attention: empty problem");
  TD_CHECK_ARG(p.Hq % p.Hkv == 0, "td_attention: Hq=%d not a multiple of Hkv=%d", p.Hq, p.Hkv);
  TD_CHECK_ARG(p.ldq % 8 == 0 && p.ldkv % 8 == 0 && p.ldo % 4 == 0, "td_attention: row strides must be 16-byte multiples");
  TD_CHECK_ARG(((long long)(p.Sq + 256) * p.ldq) * 2 < (1ll << 32) && ((long long)(p.Skv + 64) * p.ldkv) * 2 < (1ll << 32),
               "td_attention: per-batch operand exceeds the 4 GiB buffer-descriptor range");
  TD_CHECK_ARG(((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V | (uintptr_t)p.O) % 16 == 0, "td_attention: pointers must be 16-byte aligned");
  // one query token per sequence against its KV cache: the dedicated decode kernel (keys split over lanes, GQA group shares K/V)
  if (p.Sq == 1 && p.causal && !p.bias && (p.variant & 0xff) == 0 && (p.kv_lens || p.causal_offset == p.Skv - 1))
    return td_attn_decode_launch(p, stream);
  TdAttnParams q = p;
  q.q_per_kv = p.Hq / p.Hkv;
  // Structures that lost the in-process A/B on MI355X and were removed: a 4-wave workgroup with ONE wave per SIMD, each wave
  // owning two 32-row blocks half a tile apart (softmax of one block on the VALU under the other block's MFMAs, 3-deep K / V
  // rings, the guide's "512-register" shape; round 3: correct, 448 vs 219 us -- with > 256 live registers hipcc puts every
  // accumulator AND the Q fragments in AGPRs and moves ~400 values per tile through v_accvgpr_read/write, plus 1 KB of scratch),
  // ping-pong wave halves with a 3-deep V ring (-9 %),
  // 4-wave workgroups two per CU (-35 %), intra-wave QK^T(t+1) / softmax(t) software pipelining (-6 %), the first
  // lockstep kernel with per-read address arithmetic (-15 %).
  constexpr int NW = 8;
  constexpr int lds = 4 * TILE_BYTES;
  int dev = 0;
  TD_CHECK_HIP(hipGetDevice(&dev));
  static std::atomic<unsigned long long> a0{0}, a1{0}, a2{0}, a3{0}, a4{0}, a5{0};
  dim3 grid((p.Sq + NW * Q_WAVE - 1) / (NW * Q_WAVE), p.Hq, p.batch);
  if (p.bias) TD_CHECK_ARG(p.Skv % 4 == 0 && ((uintptr_t)p.bias) % 16 == 0 && p.batch == 1, "td_attention: bias needs Skv %% 4 == 0, 16-byte alignment, batch 1");
  if (p.kv_lens) TD_CHECK_ARG(p.causal && !p.bias, "td_attention: per-sequence kv lengths exist for the causal kernel only");
  if (p.q_prescaled) TD_CHECK_ARG(!p.causal && !p.bias && !p.kv_lens && !p.seg_starts, "td_attention: pre-scaled q is a form of the joint (unmasked) attention only");
  TD_CHECK_ARG(p.score_bound >= 0.f && p.score_bound <= 48.f && (p.score_bound == 0.f || p.q_prescaled), "td_attention: a score bound goes with pre-scaled q (joint attention) and lies in (0, 48] octaves (0 = none)");
  if (p.q8) TD_CHECK_ARG(!p.causal && !p.bias && !p.kv_lens && !p.seg_starts && p.batch == 1 && p.q8_inv && p.q8_amax && p.ldq8 % 8 == 0 && ((uintptr_t)p.q8) % 8 == 0,
                         "td_attention: the int8 output form is for the joint attention of one batch entry, with 8-byte aligned rows");
  if (p.seg_starts) {   // packed segments: plain grid over (query tiles of the longest segment, heads, segments)
    TD_CHECK_ARG(!p.bias && !p.kv_lens && p.Sq == p.Skv && (!p.causal || p.causal_offset == 0),
                 "td_attention(varlen): full or causal (offset 0) attention inside each segment only (no bias or cache lengths)");
    if (p.causal) {      // packed causal prefill of the Qwen2-VL engine (prompts of different lengths back to back, no padding rows)
      static std::atomic<unsigned long long> a12{0};
      if (int e = set_lds_attr_once(td_attn_fwd_d128_lean_kernel<true, NW, false, true>, lds, a12, dev)) return e;
      hipLaunchKernelGGL((td_attn_fwd_d128_lean_kernel<true, NW, false, true>), grid, dim3(NW * 64), lds, stream, q);
      TD_CHECK_LAUNCH();
      return 0;
    }
    static std::atomic<unsigned long long> a6{0};
    if (int e = set_lds_attr_once(td_attn_fwd_d128_lean_kernel<false, NW, false, true>, lds, a6, dev)) return e;
    hipLaunchKernelGGL((td_attn_fwd_d128_lean_kernel<false, NW, false, true>), grid, dim3(NW * 64), lds, stream, q);
    TD_CHECK_LAUNCH();
    return 0;
  }

  // joint attention with more workgroups than CUs: one round of persistent workgroups over equal (item, KV tile) ranges.
  // variant 1 = the one-workgroup-per-item kernel for every shape (in-process A/B); variant 2 = stream-K without the XCD remap.
  const int n_items = (int)(grid.x * grid.y * grid.z);
  const int cus = device_cus(dev);
  const int nt = (p.Skv + KV_TILE - 1) / KV_TILE;
  if (!p.causal && !p.bias && !p.kv_lens && (p.variant & 0xff) != 1 && cus > 0 && n_items > cus && cus < SK_MAX_RANGES && (long long)n_items * nt < (1ll << 31)) {
    char* ws = (char*)p.sk_ws;
    if (!ws) {
      if (int rc = sk_pooled_workspace(dev, cus, stream, &ws)) return rc;
    }
    constexpr int lds_sk = lds + 16;        // + the ticket word
    if (p.q_prescaled && p.score_bound > 0.f) {
      static std::atomic<unsigned long long> a10{0};
      if (int e = set_lds_attr_once(td_attn_fwd_d128_streamk_kernel<NW, true, true, true, true>, lds_sk, a10, dev)) return e;
      hipLaunchKernelGGL((td_attn_fwd_d128_streamk_kernel<NW, true, true, true, true>), dim3(cus), dim3(NW * 64), lds_sk, stream, q, ws, (int)grid.x, nt);
    } else if (p.q_prescaled) {
      static std::atomic<unsigned long long> a7{0};
      if (int e = set_lds_attr_once(td_attn_fwd_d128_streamk_kernel<NW, true, true>, lds_sk, a7, dev)) return e;
      hipLaunchKernelGGL((td_attn_fwd_d128_streamk_kernel<NW, true, true>), dim3(cus), dim3(NW * 64), lds_sk, stream, q, ws, (int)grid.x, nt);
    } else if ((p.variant & 0xff) == 2) {
      if (int e = set_lds_attr_once(td_attn_fwd_d128_streamk_kernel<NW, false>, lds_sk, a4, dev)) return e;
      hipLaunchKernelGGL((td_attn_fwd_d128_streamk_kernel<NW, false>), dim3(cus), dim3(NW * 64), lds_sk, stream, q, ws, (int)grid.x, nt);
    } else {
      if (int e = set_lds_attr_once(td_attn_fwd_d128_streamk_kernel<NW, true>, lds_sk, a5, dev)) return e;
      hipLaunchKernelGGL((td_attn_fwd_d128_streamk_kernel<NW, true>), dim3(cus), dim3(NW * 64), lds_sk, stream, q, ws, (int)grid.x, nt);
    }
    TD_CHECK_LAUNCH();
    return 0;
  }
  if (p.bias) {   // separate instantiation: the score-bias loads must not touch the hot no-bias instruction stream
    if (p.causal) {
      if (int e = set_lds_attr_once(td_attn_fwd_d128_lean_kernel<true, NW, true>, lds, a0, dev)) return e;
      hipLaunchKernelGGL((td_attn_fwd_d128_lean_kernel<true, NW, true>), grid, dim3(NW * 64), lds, stream, q);
    } else {
      if (int e = set_lds_attr_once(td_attn_fwd_d128_lean_kernel<false, NW, true>, lds, a1, dev)) return e;
      hipLaunchKernelGGL((td_attn_fwd_d128_lean_kernel<false, NW, true>), grid, dim3(NW * 64), lds, stream, q);
    }
  } else {
    if (p.q_prescaled && p.score_bound > 0.f) {
      static std::atomic<unsigned long long> a11{0};
      if (int e = set_lds_attr_once(td_attn_fwd_d128_lean_kernel<false, NW, false, false, true, true, true>, lds, a11, dev)) return e;
      hipLaunchKernelGGL((td_attn_fwd_d128_lean_kernel<false, NW, false, false, true, true, true>), grid, dim3(NW * 64), lds, stream, q);
    } else if (p.q_prescaled) {
      static std::atomic<unsigned long long> a8{0};
      if (int e = set_lds_attr_once(td_attn_fwd_d128_lean_kernel<false, NW, false, false, true>, lds, a8, dev)) return e;
      hipLaunchKernelGGL((td_attn_fwd_d128_lean_kernel<false, NW, false, false, true>), grid, dim3(NW * 64), lds, stream, q);
    } else if (p.causal) {
      if (int e = set_lds_attr_once(td_attn_fwd_d128_lean_kernel<true, NW>, lds, a2, dev)) return e;
      hipLaunchKernelGGL((td_attn_fwd_d128_lean_kernel<true, NW>), grid, dim3(NW * 64), lds, stream, q);
    } else {
      if (int e = set_lds_attr_once(td_attn_fwd_d128_lean_kernel<false, NW>, lds, a3, dev)) return e;
      hipLaunchKernelGGL((td_attn_fwd_d128_lean_kernel<false, NW>), grid, dim3(NW * 64), lds, stream, q);
    }
  }
  TD_CHECK_LAUNCH();
  return 0;
}

// bytes of the stream-K hand-off workspace td_attn_launch wants in TdAttnParams::sk_ws on the current device (zero-filled
// once by the owner; one per concurrently running launch, e.g. one per engine context)
size_t td_attn_streamk_ws_bytes() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  const int cus = device_cus(dev);
  return cus > 0 && cus < SK_MAX_RANGES ? sk_ws_bytes(cus) : 0;
}
