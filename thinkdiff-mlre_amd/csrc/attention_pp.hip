// Joint attention, head_dim 128, for gfx950 -- the 4-wave "two row blocks per wave" form (round 3).
//
// The 8-wave kernels of attention_bf16.hip put two waves on every SIMD and let the hardware overlap one wave's softmax (VALU)
// with the other's MFMAs.  rocprofv3 SQ counters said what that costs: neither pipe is busy (matrix 62-74 %, LDS ~65 %), the
// two waves of a SIMD pace each other through the issue port, and every wave reads each K / V fragment from LDS for 32 query
// rows only.  This kernel is the other classic shape (guide: "one wave per SIMD, 512-register kernel"):
//  * workgroup = 4 waves, ONE per SIMD, 512 registers each; a wave owns 64 query rows as two 32-row blocks A and B;
//  * inside a wave the two blocks run half a tile apart: while the matrix pipe works on one block (its P.V of the tile just
//    exponentiated and the QK^T of its next tile) the wave's VALU does the other block's softmax -- the MFMA is asynchronous to
//    the VALU of its own wave, so the overlap needs no second wave and no arbitration:
//        X(t):  VALU softmax_A(t)   ||  MFMA  P.V_B(t-1), QK^T_B(t)
//        Y(t):  VALU softmax_B(t)   ||  MFMA  P.V_A(t),   QK^T_A(t+1)
//  * K(t), K(t+1), V(t-1), V(t) are live inside one iteration, so K and V each get a 3-deep LDS ring (96 KiB), filled by LDS-DMA
//    two tiles ahead; ONE workgroup barrier per 64-key tile, for 4 waves instead of 8;
//  * same (item, KV tile) stream-K decomposition, hand-off protocol and workspace as td_attn_fwd_d128_streamk_kernel -- a wave's
//    two blocks take the slots of two of that kernel's waves -- and the same tile math (reference point and row sums on the
//    matrix pipe, q pre-scaled: attention_common.h).  With gridDim.x == number of items every range is exactly one item: the
//    plain-grid form, no hand-off.
#include "attention_common.h"

template <bool PRE>
__global__ __launch_bounds__(256, 1) void td_attn_fwd_d128_pp_kernel(const TdAttnParams p, char* __restrict__ ws, const int n_qblk, const int nt) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [K ring: 3 x 16 KiB | V ring: 3 x 16 KiB | ticket word]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5;
  const int l31 = lane & 31;
  unsigned* const cnt = (unsigned*)ws + 16;
  TD_LDS unsigned* const ticket_lds = (TD_LDS unsigned*)(smem + 6 * TILE_BYTES);

  const int G = gridDim.x;
  int r = blockIdx.x;
  {   // workgroups of one XCD take neighbouring ranges (= neighbouring query tiles of the same heads: the same K/V in that L2)
    const int q8 = G >> 3, r8 = G & 7, xcd = r & 7;
    r = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (r >> 3);
  }
  const long long total = (long long)n_qblk * p.Hq * p.batch * nt;
  long long it = total * r / G;
  const long long it_end = total * (r + 1) / G;

  const int Skv = p.Skv;
  constexpr int GROUPS = KV_TILE / 4;          // 16 DMA groups of 4 rows per tile
  constexpr int SG = GROUPS / NW;              // 4 per wave
  const int srow = lane >> 4;
  const float c = p.scale * 1.4426950408889634f;
  const float cc = PRE ? 1.0f : c;
  const bf16x8_t kone = {(short)(h5 == 0 ? 0x3F80 : 0), 0, 0, 0, 0, 0, 0, 0};

  // per-lane LDS byte addresses inside a tile (ring slot added per use)
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
        va[jj][db] = lds0 + 3 * TILE_BYTES + (4 * h5 + vq + 8 * jj) * 256 + (((4 * db + vchunk) ^ swz) << 4) + 8 * (vp & 1);
    }
  }
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bf16x8_t ones8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

  while (it < it_end) {
    const int item = (int)(it / nt);
    const int kb = (int)(it - (long long)item * nt);
    const int ke = (int)min((long long)nt, kb + (it_end - it));
    __builtin_assume(ke > kb);
    it += ke - kb;
    const int qblk = item % n_qblk;
    const int hb = item / n_qblk;
    const int head = hb % p.Hq, batch = hb / p.Hq;
    const int kvhead = head / p.q_per_kv;
    const int q0 = qblk * 256 + wid * 64;          // block A: rows q0 .. q0+31, block B: q0+32 .. q0+63

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
      const int g = wid + NW * s;
      const int swz = (srow << 2) | (g & 3);
      const int chunk = (lane & 15) ^ swz;
      voffK[s] = (unsigned)(g * 4 + srow) * (unsigned)p.ldkv * 2u + (unsigned)(kvhead * D + chunk * 8) * 2u;
    }
    auto stageK = [&](int t) {      // tile t -> K ring slot t % 3
      const unsigned tile_off = (unsigned)t * KV_TILE * (unsigned)p.ldkv * 2u;
      char* dst = smem + (t % 3) * TILE_BYTES;
#pragma unroll
      for (int s = 0; s < SG; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (TD_LDS void*)(dst + (wid + NW * s) * 1024), 16, voffK[s] + tile_off, 0, 0, 0);
    };
    auto stageV = [&](int t) {
      const unsigned tile_off = (unsigned)t * KV_TILE * (unsigned)p.ldkv * 2u;
      char* dst = smem + (3 + t % 3) * TILE_BYTES;
#pragma unroll
      for (int s = 0; s < SG; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (TD_LDS void*)(dst + (wid + NW * s) * 1024), 16, voffK[s] + tile_off, 0, 0, 0);
    };

    __syncthreads();          // the previous part's tiles are no longer read by any wave
    stageK(kb);
    stageV(kb);
    if (kb + 1 < ke) stageK(kb + 1);

    bf16x8_t qf[2][8];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const unsigned qoff = (unsigned)(q0 + 32 * b + l31) * (unsigned)p.ldq * 2u + (unsigned)(head * D + 8 * h5) * 2u;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsQ, qoff + ks * 32, 0, 0);
        qf[b][ks] = __builtin_bit_cast(bf16x8_t, v);
      }
    }

    f32x16_t o[2][4], lacc[2], st[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) o[b][db][rr] = 0.f;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) lacc[b][rr] = 0.f;
    }
    float m_run[2] = {0.f, 0.f};
    bf16x8_t qnegm[2] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
    bf16x8_t pf[2][2][2];       // [block][kb][s]: the probabilities of the block's current tile, as P.V operands

    // ---- building blocks.  Each takes the tile index for its LDS ring slot; MFMA groups are emitted in pieces so that the
    // caller can thread VALU work of the other block between them ---------------------------------------------------------
    auto kfrag = [&](int t, int e) {   // e = kb * 8 + ks
      return *(const TD_LDS bf16x8_t*)(uintptr_t)(ka[e & 7] + (unsigned)(t % 3) * TILE_BYTES + (e >> 3) * 32 * 256);
    };
    auto vfrag = [&](int t, int e) {   // e = (kb * 2 + s) * 4 + db
      const unsigned so = (unsigned)(t % 3) * TILE_BYTES + (e >> 2) * 16 * 256;
      const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TD_LDS bf16x4_t*)(uintptr_t)(va[0][e & 3] + so));
      const bf16x4_t b2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TD_LDS bf16x4_t*)(uintptr_t)(va[1][e & 3] + so));
      bf16x8_t vf;
      vf[0] = a[0]; vf[1] = a[1]; vf[2] = a[2]; vf[3] = a[3];
      vf[4] = b2[0]; vf[5] = b2[1]; vf[6] = b2[2]; vf[7] = b2[3];
      return vf;
    };
    // QK^T of block b, tile t, k-steps [e0, e1) of the 16 (two 32-key halves x 8 k-steps).  K fragments are read two MFMAs ahead
    // of their use and pinned there (unpinned, hipcc hoists every read of a group to its top and the kernel spills).
    auto qk_part = [&](auto b_tag, int t, auto e0_tag, auto e1_tag) {
      constexpr int b = decltype(b_tag)::value, e0 = decltype(e0_tag)::value, e1 = decltype(e1_tag)::value;
      constexpr int KPF = 2;
      bf16x8_t kf[16];
#pragma unroll
      for (int e = e0; e < e0 + KPF && e < e1; ++e) kf[e] = kfrag(t, e);
      __builtin_amdgcn_sched_group_barrier(0x100, (e1 - e0) < KPF ? (e1 - e0) : KPF, 0);
#pragma unroll
      for (int e = e0; e < e1; ++e) {
        if ((e & 7) == 0) {
          st[b][e >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kone, qnegm[b], zero16, 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if (e + KPF < e1) kf[e + KPF] = kfrag(t, e + KPF);
        st[b][e >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[e], qf[b][e & 7], st[b][e >> 3], 0, 0, 0);
        if (e + KPF < e1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    };
    // P.V of block b, tile t, steps [e0, e1) of the 16 ((kb, s) x 4 d-blocks), row sums riding on every fourth
    auto pv_part = [&](auto b_tag, int t, auto e0_tag, auto e1_tag) {
      constexpr int b = decltype(b_tag)::value, e0 = decltype(e0_tag)::value, e1 = decltype(e1_tag)::value;
      constexpr int VPF = 2;
      bf16x8_t vf[16];
#pragma unroll
      for (int e = e0; e < e0 + VPF && e < e1; ++e) vf[e] = vfrag(t, e);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * ((e1 - e0) < VPF ? (e1 - e0) : VPF), 0);
#pragma unroll
      for (int e = e0; e < e1; ++e) {
        if (e + VPF < e1) vf[e + VPF] = vfrag(t, e + VPF);
        o[b][e & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[e], pf[b][e >> 3][(e >> 2) & 1], o[b][e & 3], 0, 0, 0);
        if (e + VPF < e1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if ((e & 3) == 3) {
          lacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pf[b][e >> 3][(e >> 2) & 1], lacc[b], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      }
    };
    // keys past Skv of the last tile
    auto mask_tail = [&](auto b_tag, int t) {
      constexpr int b = decltype(b_tag)::value;
      const int key0 = t * KV_TILE;
      if (key0 + KV_TILE > Skv) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            const int key = key0 + kk * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h5;
            if (key >= Skv) st[b][kk][rr] = -INFINITY;
          }
      }
    };
    // softmax, first half: row maximum of the fresh scores and (rarely) a move of the reference point
    auto sm_max = [&](auto b_tag, bool first) {
      constexpr int b = decltype(b_tag)::value;
      float mx = st[b][0][0];
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) mx = max3(mx, st[b][0][rr], st[b][1][rr]);
      mx = half_swap_max(mx);
      constexpr float RESCALE_LOG2 = 8.0f;
      if (first || __any(mx * cc > RESCALE_LOG2)) {
        float target = first ? mx : fmaxf(mx, 0.f);
        if (!(target > -INFINITY)) target = 0.f;
        const float m_new = bf16_ceil(m_run[b] + target);
        const float d = m_new - m_run[b];
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-d * cc);
#pragma unroll
          for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) o[b][db][rr] *= alpha;
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) lacc[b][rr] *= alpha;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) st[b][kk][rr] -= d;
        m_run[b] = m_new;
        qnegm[b][0] = h5 == 0 ? (short)f2bf(-m_new) : (short)0;
      }
    };
    // softmax, second half: exponentials of quarter `part` (0..3) of the tile -> one P fragment
    auto sm_exp = [&](auto b_tag, auto part_tag) {
      constexpr int b = decltype(b_tag)::value, part = decltype(part_tag)::value;
      constexpr int kk = part >> 1, s = part & 1;
      u32x4_t pk;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = st[b][kk][8 * s + 2 * j], x1 = st[b][kk][8 * s + 2 * j + 1];
        pk[j] = pack_bf2(__builtin_amdgcn_exp2f(PRE ? x0 : x0 * c), __builtin_amdgcn_exp2f(PRE ? x1 : x1 * c));
      }
      pf[b][kk][s] = __builtin_bit_cast(bf16x8_t, pk);
    };
    // One half-iteration: softmax of block `bs` (tile ts) on the VALU, threaded through the MFMAs of block `bm`:
    // P.V(bm, tpv) if HAS_PV, then QK^T(bm, tqk) if HAS_QK.
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I6 = std::integral_constant<int, 6>; using I8 = std::integral_constant<int, 8>;
    using I12 = std::integral_constant<int, 12>; using I16 = std::integral_constant<int, 16>;
    auto segment = [&](auto has_pv, auto has_qk, auto bs, int ts, bool first, auto bm, int tpv, int tqk) {
      constexpr bool HAS_PV = decltype(has_pv)::value, HAS_QK = decltype(has_qk)::value;
      mask_tail(bs, ts);
      if constexpr (HAS_PV) pv_part(bm, tpv, I0{}, I6{});
      sm_max(bs, first);
      if constexpr (HAS_PV) { pv_part(bm, tpv, I6{}, I8{}); sm_exp(bs, I0{}); pv_part(bm, tpv, I8{}, I12{}); sm_exp(bs, I1{}); pv_part(bm, tpv, I12{}, I16{}); }
      else { sm_exp(bs, I0{}); sm_exp(bs, I1{}); }
      if constexpr (HAS_QK) { qk_part(bm, tqk, I0{}, I6{}); sm_exp(bs, I2{}); qk_part(bm, tqk, I6{}, I12{}); sm_exp(bs, I3{}); qk_part(bm, tqk, I12{}, I16{}); }
      else { sm_exp(bs, I2{}); sm_exp(bs, I3{}); }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;

    // ---- prologue: the first tile of the part (plain order: its cost is 1 tile in 68 - 108) --------------------------------
    __syncthreads();          // K(kb), V(kb), K(kb+1) landed (vmcnt(0) inside)
    if (kb + 1 < ke) stageV(kb + 1);
    if (kb + 2 < ke) stageK(kb + 2);
    qk_part(I0{}, kb, I0{}, I16{});
    qk_part(I1{}, kb, I0{}, I16{});
    segment(F_{}, F_{}, I0{}, kb, true, I1{}, 0, 0);                             // softmax_A(kb)
    // Y(kb): softmax_B(kb) || P.V_A(kb), QK^T_A(kb+1).  In the LAST tile's Y the QK^T runs on the last tile again (its scores are
    // never used): one instantiation of every segment instead of one per edge case keeps the loop body small.
    segment(T_{}, T_{}, I1{}, kb, true, I0{}, kb, min(kb + 1, ke - 1));
    // ---- steady state ------------------------------------------------------------------------------------------------------
    for (int t = kb + 1; t < ke; ++t) {
      __syncthreads();        // K(t+1), V(t) landed; ring slots of K(t-1), V(t-2) are free
      if (t + 2 < ke) stageK(t + 2);
      if (t + 1 < ke) stageV(t + 1);
      segment(T_{}, T_{}, I0{}, t, false, I1{}, t - 1, t);                       // X(t): softmax_A(t) || P.V_B(t-1), QK^T_B(t)
      segment(T_{}, T_{}, I1{}, t, false, I0{}, t, min(t + 1, ke - 1));          // Y(t): softmax_B(t) || P.V_A(t), QK^T_A(t+1)
    }
    pv_part(I1{}, ke - 1, I0{}, I16{});                                          // epilogue: P.V_B of the last tile

    // ---- hand-off / finish, per block (a block takes the slot of one wave of the 8-wave kernel: 2 wid + b) --------------------
    float l_run[2] = {lacc[0][0], lacc[1][0]};
    if (kb > 0 || ke < nt) {
      const int j = kb > 0 ? r : r + 1;
      const bool tail = kb > 0;
      const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + SK_HEADER_BYTES), 0, (unsigned)(2 * G) * SK_SLOT_FLOATS * 4u, 0x00020000);
      auto ask = [&](bool draw) -> unsigned {
        __syncthreads();
        if (tid == 0)
          *ticket_lds = draw ? __hip_atomic_fetch_add(cnt + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : __hip_atomic_load(cnt + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        return *ticket_lds;
      };
      bool other_ready = !tail && ask(false) == 1u;
      if (!other_ready) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const unsigned lane_off = ((unsigned)(2 * wid + b) * 17u * 64u + (unsigned)lane) * 16u;
          const unsigned mine = (unsigned)(tail ? j : G + j) * (SK_SLOT_FLOATS * 4u) + lane_off;
#pragma unroll
          for (int q4 = 0; q4 < 16; ++q4) {
            const u32x4_t v = {as_u32(o[b][q4 >> 2][4 * (q4 & 3)]), as_u32(o[b][q4 >> 2][4 * (q4 & 3) + 1]),
                               as_u32(o[b][q4 >> 2][4 * (q4 & 3) + 2]), as_u32(o[b][q4 >> 2][4 * (q4 & 3) + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, rsS, mine + q4 * 1024u, 0, 16);
          }
          const u32x4_t ml = {as_u32(m_run[b]), as_u32(l_run[b]), 0u, 0u};
          __builtin_amdgcn_raw_buffer_store_b128(ml, rsS, mine + 16 * 1024u, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        other_ready = ask(true) == 1u;
        if (!other_ready) continue;
      }
      if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const unsigned lane_off = ((unsigned)(2 * wid + b) * 17u * 64u + (unsigned)lane) * 16u;
        const unsigned theirs = (unsigned)(tail ? G + j : j) * (SK_SLOT_FLOATS * 4u) + lane_off;
        const u32x4_t ml = __builtin_amdgcn_raw_buffer_load_b128(rsS, theirs + 16 * 1024u, 0, 0);
        const float m2 = as_f32(ml[0]), l2 = as_f32(ml[1]);
        const float mm = fmaxf(m_run[b], m2);
        const float a1 = __builtin_amdgcn_exp2f((m_run[b] - mm) * cc), a2 = __builtin_amdgcn_exp2f((m2 - mm) * cc);
        auto comb = [&](float own, float other) {
          return tail ? __builtin_fmaf(other, a2, own * a1) : __builtin_fmaf(own, a1, other * a2);
        };
        l_run[b] = comb(l_run[b], l2);
#pragma unroll
        for (int q4 = 0; q4 < 16; ++q4) {
          const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsS, theirs + q4 * 1024u, 0, 0);
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4)
            o[b][q4 >> 2][4 * (q4 & 3) + k4] = comb(o[b][q4 >> 2][4 * (q4 & 3) + k4], as_f32(v[k4]));
        }
      }
      if (tid == 0) __hip_atomic_store(cnt + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = q0 + 32 * b + l31;
      attn_store_rows(o[b], 1.0f / l_run[b], p.O + (size_t)batch * p.o_bstride + (size_t)min(q, p.Sq - 1) * p.ldo + head * D, h5,
                      (p.ldo & 7) == 0, q < p.Sq);
    }
  }
#endif
}

int td_attn_pp_launch(const TdAttnParams& q, char* ws, int grid, int n_qblk, int nt, hipStream_t stream) {
  constexpr int lds = 6 * TILE_BYTES + 16;
  static std::atomic<unsigned long long> a0{0}, a1{0};
  int dev = 0;
  TD_CHECK_HIP(hipGetDevice(&dev));
  auto once = [&](const void* fn, std::atomic<unsigned long long>& done) -> int {
    if (!((done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
      TD_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    return 0;
  };
  if (q.q_prescaled) {
    if (int e = once((const void*)td_attn_fwd_d128_pp_kernel<true>, a0)) return e;
    hipLaunchKernelGGL((td_attn_fwd_d128_pp_kernel<true>), dim3(grid), dim3(256), lds, stream, q, ws, n_qblk, nt);
  } else {
    if (int e = once((const void*)td_attn_fwd_d128_pp_kernel<false>, a1)) return e;
    hipLaunchKernelGGL((td_attn_fwd_d128_pp_kernel<false>), dim3(grid), dim3(256), lds, stream, q, ws, n_qblk, nt);
  }
  TD_CHECK_LAUNCH();
  return 0;
}
