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
#include <type_traits>

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

}  // namespace

template <bool CAUSAL, int NWAVES, bool STAGGER>
__global__ __launch_bounds__(NWAVES * 64, 2) void td_attn_fwd_d128_kernel(const TdAttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int VBUFS = STAGGER ? 3 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [K tile x 2 | V tile x VBUFS]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5;       // lane half
  const int l31 = lane & 31;

  const int qblk = blockIdx.x;
  const int head = blockIdx.y;
  const int batch = blockIdx.z;
  const int kvhead = head / p.q_per_kv;
  const int q0 = qblk * (NWAVES * Q_WAVE) + wid * Q_WAVE;  // first query row of this wave

  const bf16_t* Qb = p.Q + (size_t)batch * p.q_bstride;
  const bf16_t* Kb = p.K + (size_t)batch * p.kv_bstride;
  const bf16_t* Vb = p.V + (size_t)batch * p.kv_bstride;
  bf16_t* Ob = p.O + (size_t)batch * p.o_bstride;

  const unsigned q_bytes = (unsigned)(((long long)(p.Sq - 1) * p.ldq + p.Hq * D) * 2);
  const unsigned kv_bytes = (unsigned)(((long long)(p.Skv - 1) * p.ldkv + p.Hkv * D) * 2);
  __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qb, 0, q_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, kv_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, kv_bytes, 0x00020000);

  // ---- number of KV tiles this workgroup visits ----------------------------------------------
  int nt = (p.Skv + KV_TILE - 1) / KV_TILE;
  if (CAUSAL) {
    const int last_q = min(p.Sq, (qblk + 1) * NWAVES * Q_WAVE) - 1 + p.causal_offset;
    nt = min(nt, last_q / KV_TILE + 1);
  }

  // ---- staging (LDS-DMA): 16 groups of 4 rows per tile, NWAVES waves share them ---------------
  constexpr int GROUPS = KV_TILE / 4;
  constexpr int SG = (GROUPS + NWAVES - 1) / NWAVES;
  const int srow = lane >> 4;  // row inside the 4-row group
  unsigned voffK[SG];
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    const int g = wid + NWAVES * s;
    const int swz = (srow << 2) | (g & 3);
    const int chunk = (lane & 15) ^ swz;
    voffK[s] = (unsigned)(g * 4 + srow) * (unsigned)p.ldkv * 2u + (unsigned)(kvhead * D + chunk * 8) * 2u;
  }
  auto stage = [&](int kslot, int vslot_, int t) {
    char* kdst = smem + kslot * TILE_BYTES;
    char* vdst = smem + 2 * TILE_BYTES + vslot_ * TILE_BYTES;
    const unsigned tile_off = (unsigned)t * KV_TILE * (unsigned)p.ldkv * 2u;
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const int g = wid + NWAVES * s;
      if (GROUPS % NWAVES == 0 || g < GROUPS) {
        // row part stays in voffset so the descriptor range check sees it: keys >= Skv read as 0
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (TD_LDS void*)(kdst + g * 1024), 16, voffK[s] + tile_off, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (TD_LDS void*)(vdst + g * 1024), 16, voffK[s] + tile_off, 0, 0, 0);
      }
    }
  };

  stage(0, 0, 0);

  // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q0 + l31][16 ks + 8 h5 .. +8] -----
  bf16x8_t qf[8];
  {
    const unsigned qoff = (unsigned)(q0 + l31) * (unsigned)p.ldq * 2u + (unsigned)(head * D + 8 * h5) * 2u;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsQ, qoff + ks * 32, 0, 0);
      qf[ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }

  // ---- per-lane LDS read offsets ----------------------------------------------------------------
  // K row read: row = kb*32 + l31, chunk = 2 ks + h5
  const int ksw = ((lane & 3) << 2) | ((lane >> 2) & 3);
  const int krow_off = l31 * 256;
  // V transposed read: lane i=lane&15 of its 16-lane group addresses row q=i>>2, cols 4p..4p+3 (p=i&3)
  const int vq = (lane & 15) >> 2, vp = lane & 3;
  const int vrow_base = 4 * h5 + vq;                 // + kb*32 + 16 s + 8 jj
  const int vchunk_base = 2 * ((lane >> 4) & 1) + (vp >> 1);  // + 4 db

  f32x16_t o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -1e30f;   // row max the exponentials are taken against (raw score units)
  float l_run = 0.f;      // this lane-half's partial row sum
  const float c = p.scale * 1.4426950408889634f;  // exp(x*scale) = exp2(x*c)
  const int q_pos = q0 + l31 + p.causal_offset;   // causal: keys <= q_pos visible
  bf16x8_t pf[2][2];      // [kb][s] B-operand fragments of P^T (live across the barrier for the deferred half)

  // O^T += V^T . P^T for one KV tile
  auto pv = [&](const char* vbuf) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          bf16x4_t v01[2];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int row = kb * 32 + 16 * s + 8 * jj + vrow_base;
            const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
            const int off = row * 256 + (((4 * db + vchunk_base) ^ swz) << 4) + 8 * (vp & 1);
            v01[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TD_LDS bf16x4_t*)(vbuf + off));
          }
          bf16x8_t vf;
          vf[0] = v01[0][0]; vf[1] = v01[0][1]; vf[2] = v01[0][2]; vf[3] = v01[0][3];
          vf[4] = v01[1][0]; vf[5] = v01[1][1]; vf[6] = v01[1][2]; vf[7] = v01[1][3];
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][s], o[db], 0, 0, 0);
        }
      }
    }
  };

  // Ping-pong: the two waves that share a SIMD (w and w + NWAVES/2) run half a tile apart.  The second
  // half of the workgroup defers each tile's P.V product until after the next barrier, so its matrix work
  // overlaps the first half's softmax (VALU) and vice versa; in lockstep both waves would sit in their
  // softmax at the same time with the matrix pipe idle.  The deferred product still reads V(t-1) while
  // tile t+1 is being staged, hence the 3-deep V ring (K stays 2-deep).
  const bool deferred = STAGGER && wid >= NWAVES / 2;
  int vslot = 0;  // t % VBUFS

  for (int t = 0; t < nt; ++t) {
    __syncthreads();  // vmcnt(0) + barrier: tile t landed; K buffer (t+1)&1 and V slot (t+1)%VBUFS are free
    const int vnext = vslot + 1 == VBUFS ? 0 : vslot + 1;
    if (t + 1 < nt) stage((t + 1) & 1, vnext, t + 1);
    const char* kbuf = smem + (t & 1) * TILE_BYTES;
    const char* vbuf = smem + 2 * TILE_BYTES + vslot * TILE_BYTES;
    if (deferred && t > 0) pv(smem + 2 * TILE_BYTES + (vslot == 0 ? VBUFS - 1 : vslot - 1) * TILE_BYTES);
    vslot = vnext;

    // ---- S^T = K . Q^T ------------------------------------------------------------------------
    f32x16_t st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8_t kf = *(const bf16x8_t*)(kbuf + kb * 32 * 256 + krow_off + (((2 * ks + h5) ^ ksw) << 4));
        st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[kb], 0, 0, 0);
      }
    }

    // ---- masking: tail keys (>= Skv) and causal --------------------------------------------------
    const int key0 = t * KV_TILE;
    const bool need_mask = (key0 + KV_TILE > p.Skv) || (CAUSAL && key0 + KV_TILE - 1 > q0 + p.causal_offset);
    if (need_mask) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h5;
          const bool dead = (key >= p.Skv) || (CAUSAL && key > q_pos);
          if (dead) st[kb][r] = -INFINITY;
        }
    }

    // ---- online softmax (row = query = lane & 31, split over the two lane halves) -------------
    float mx = st[0][0];
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = max3(mx, st[0][r], st[1][r]);
    mx = half_swap_max(mx);
    // Deferred rescale: O and l are rescaled only when some row's max grew by more than 2^RESCALE_LOG2
    // (P <= 2^8 then, harmless in the fp32 accumulators and in bf16 P); rows are otherwise exponentiated
    // against their stale max.  Every row is finite after tile 0: m_run starts at -1e30 so tile 0 rescales.
    constexpr float RESCALE_LOG2 = 8.0f;
    if (__any((mx - m_run) * c > RESCALE_LOG2)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    const float mc = m_run * c;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4_t pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p0 = __builtin_amdgcn_exp2f(st[kb][8 * s + 2 * j] * c - mc);
          const float p1 = __builtin_amdgcn_exp2f(st[kb][8 * s + 2 * j + 1] * c - mc);
          psum += p0 + p1;
          pk[j] = pack_bf2(p0, p1);
        }
        pf[kb][s] = __builtin_bit_cast(bf16x8_t, pk);
      }
    }
    l_run += psum;

    if (!deferred) pv(vbuf);
  }
  if (deferred) pv(smem + 2 * TILE_BYTES + (vslot == 0 ? VBUFS - 1 : vslot - 1) * TILE_BYTES);

  // ---- normalise and store: lane holds O[q][db*32 + (r&3) + 8 (r>>2) + 4 h5] --------------------
  const float l_tot = half_swap_sum(l_run);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < p.Sq) {
    bf16_t* op = Ob + (size_t)q * p.ldo + head * D + 4 * h5;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t w;
        w[0] = pack_bf2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
        w[1] = pack_bf2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        *(u32x2_t*)(op + db * 32 + 8 * g) = w;
      }
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// Lockstep structure with a lean instruction stream.  rocprofv3 PMC on the kernel above (S=4289, 24 heads):
// 7.2 VALU instructions per MFMA and the SIMD's issue slots, not the matrix pipe (50 % busy), set the pace --
// two thirds of that VALU work was LDS address arithmetic (XOR-swizzled addresses recomputed per read) and
// accumulator zeroing.  Here every per-lane LDS address lives in a register for the whole kernel (8 for the
// K row reads, 8 for the V transposed reads; k-block / k-step / tile-slot parts are instruction immediates
// and one XOR per register per tile flips the double-buffer slot).
// ---------------------------------------------------------------------------------------------
template <bool CAUSAL, int NWAVES, bool BIAS = false>
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

  const unsigned q_bytes = (unsigned)(((long long)(p.Sq - 1) * p.ldq + p.Hq * D) * 2);
  // batched KV-cached decode: sequence `batch` has its own cache length (causal instantiation only, so the joint-attention
  // instruction stream of FLUX is untouched); keys visible to query row q: key <= q + (Skv - Sq)
  int Skv = p.Skv, c_off = p.causal_offset;
  if constexpr (CAUSAL) {
    if (p.kv_lens) { Skv = p.kv_lens[batch]; c_off = Skv - p.Sq; }
  }
  const unsigned kv_bytes = (unsigned)(((long long)(Skv - 1) * p.ldkv + p.Hkv * D) * 2);
  __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qb, 0, q_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, kv_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, kv_bytes, 0x00020000);

  int nt = (Skv + KV_TILE - 1) / KV_TILE;
  if (CAUSAL) {
    const int last_q = min(p.Sq, (qblk + 1) * NWAVES * Q_WAVE) - 1 + c_off;
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
  float m_run = -1e30f;
  float l_run = 0.f;
  const float c = p.scale * 1.4426950408889634f;
  const int q_pos = q0 + l31 + c_off;
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // The tile body is instantiated for both double-buffer slots: the slot offset of every LDS read is then an instruction
  // immediate instead of 16 address flips (v_xor) per tile and wave.
  auto tile = [&](const int t, auto slot_tag) {
    constexpr unsigned PO = decltype(slot_tag)::value * TILE_BYTES;
    __syncthreads();  // vmcnt(0) + barrier: tile t landed; slot (t+1)&1 no longer read
    if (t + 1 < nt) stage((t + 1) & 1, t + 1);

    // ---- S^T = K . Q^T ------------------------------------------------------------------------
    // K fragments are fetched KPF MFMAs ahead of their use (pinned: hipcc would issue each read right
    // before its consumer and expose the LDS latency 16 times per tile)
    f32x16_t st[2];
    {
      constexpr int KPF = 2;   // depth re-measured in-process at S = 4289: 2 / 2 beats 4 / 3 and 6 / 5 by 2 % (1 / 1 ties)
      auto kread = [&](int e) {   // e = kb * 8 + ks
        return *(const TD_LDS bf16x8_t*)(uintptr_t)(ka[e & 7] + PO + (e >> 3) * 32 * 256);
      };
      bf16x8_t kf[16];
#pragma unroll
      for (int e = 0; e < KPF; ++e) kf[e] = kread(e);
      __builtin_amdgcn_sched_group_barrier(0x100, KPF, 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (e + KPF < 16) kf[e + KPF] = kread(e + KPF);
        st[e >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[e], qf[e & 7], (e & 7) == 0 ? zero16 : st[e >> 3], 0, 0, 0);
        if (e + KPF < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }

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

    // ---- online softmax with deferred rescale -----------------------------------------------------
    float mx = st[0][0];
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = max3(mx, st[0][r], st[1][r]);
    mx = half_swap_max(mx);
    constexpr float RESCALE_LOG2 = 8.0f;
    if (__any((mx - m_run) * c > RESCALE_LOG2)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    const float mc = m_run * c;
    float psum = 0.f;
    bf16x8_t pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4_t pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p0 = __builtin_amdgcn_exp2f(st[kb][8 * s + 2 * j] * c - mc);
          const float p1 = __builtin_amdgcn_exp2f(st[kb][8 * s + 2 * j + 1] * c - mc);
          psum += p0 + p1;
          pk[j] = pack_bf2(p0, p1);
        }
        pf[kb][s] = __builtin_bit_cast(bf16x8_t, pk);
      }
    }
    l_run += psum;

    // ---- O^T += V^T . P^T ------------------------------------------------------------------------
    {
      constexpr int VPF = 2;   // V^T fragments in flight ahead of their MFMA (2 transposed reads each)
      auto vread = [&](int e, int jj) {   // e = (kb * 2 + s) * 4 + db
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((TD_LDS bf16x4_t*)(uintptr_t)(va[jj][e & 3] + PO + (e >> 2) * 16 * 256));
      };
      bf16x4_t v0[16], v1[16];
#pragma unroll
      for (int e = 0; e < VPF; ++e) { v0[e] = vread(e, 0); v1[e] = vread(e, 1); }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * VPF, 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (e + VPF < 16) { v0[e + VPF] = vread(e + VPF, 0); v1[e + VPF] = vread(e + VPF, 1); }
        bf16x8_t vf;
        vf[0] = v0[e][0]; vf[1] = v0[e][1]; vf[2] = v0[e][2]; vf[3] = v0[e][3];
        vf[4] = v1[e][0]; vf[5] = v1[e][1]; vf[6] = v1[e][2]; vf[7] = v1[e][3];
        o[e & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[e >> 3][(e >> 2) & 1], o[e & 3], 0, 0, 0);
        if (e + VPF < 16) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }

  };
  {
    int t = 0;
    for (; t + 1 < nt; t += 2) {
      tile(t, std::integral_constant<unsigned, 0>{});
      tile(t + 1, std::integral_constant<unsigned, 1>{});
    }
    if (t < nt) tile(t, std::integral_constant<unsigned, 0>{});
  }

  const float l_tot = half_swap_sum(l_run);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < p.Sq) {
    bf16_t* op = Ob + (size_t)q * p.ldo + head * D + 4 * h5;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t w;
        w[0] = pack_bf2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
        w[1] = pack_bf2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        *(u32x2_t*)(op + db * 32 + 8 * g) = w;
      }
  }
#endif
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
  if (p.Sq == 1 && p.causal && !p.bias && p.variant == 0 && (p.kv_lens || p.causal_offset == p.Skv - 1))
    return td_attn_decode_launch(p, stream);
  TdAttnParams q = p;
  q.q_per_kv = p.Hq / p.Hkv;
  // variant 0 (shipped): lean instruction stream (resident LDS addresses, prefetched fragments);
  // variant 1: the first lockstep kernel, kept for in-process A/B.  Structures that lost the A/B on MI355X
  // and were removed: ping-pong wave halves with a 3-deep V ring (-9 %), 4-wave workgroups two per CU (-35 %),
  // intra-wave QK^T(t+1) / softmax(t) software pipelining on the fat instruction stream (-6 %).
  constexpr int NW = 8;
  static bool attr_set = false;
  if (!attr_set) {
    TD_CHECK_HIP(hipFuncSetAttribute((const void*)td_attn_fwd_d128_kernel<false, NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES));
    TD_CHECK_HIP(hipFuncSetAttribute((const void*)td_attn_fwd_d128_kernel<true, NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES));
    TD_CHECK_HIP(hipFuncSetAttribute((const void*)td_attn_fwd_d128_lean_kernel<false, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES));
    TD_CHECK_HIP(hipFuncSetAttribute((const void*)td_attn_fwd_d128_lean_kernel<true, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES));
    TD_CHECK_HIP(hipFuncSetAttribute((const void*)td_attn_fwd_d128_lean_kernel<false, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES));
    TD_CHECK_HIP(hipFuncSetAttribute((const void*)td_attn_fwd_d128_lean_kernel<true, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES));
    attr_set = true;
  }
  dim3 grid((p.Sq + NW * Q_WAVE - 1) / (NW * Q_WAVE), p.Hq, p.batch);
  const int lds = 4 * TILE_BYTES;
  if (p.bias) TD_CHECK_ARG(p.Skv % 4 == 0 && ((uintptr_t)p.bias) % 16 == 0 && p.batch == 1, "td_attention: bias needs Skv %% 4 == 0, 16-byte alignment, batch 1");
  if (p.kv_lens) TD_CHECK_ARG(p.causal && !p.bias, "td_attention: per-sequence kv lengths exist for the causal kernel only");
  if (p.variant == 1 && !p.bias && !p.kv_lens) {
    if (p.causal) hipLaunchKernelGGL((td_attn_fwd_d128_kernel<true, NW, false>), grid, dim3(NW * 64), lds, stream, q);
    else hipLaunchKernelGGL((td_attn_fwd_d128_kernel<false, NW, false>), grid, dim3(NW * 64), lds, stream, q);
  } else if (p.bias) {   // separate instantiation: the score-bias loads must not touch the hot no-bias instruction stream
    if (p.causal) hipLaunchKernelGGL((td_attn_fwd_d128_lean_kernel<true, NW, true>), grid, dim3(NW * 64), lds, stream, q);
    else hipLaunchKernelGGL((td_attn_fwd_d128_lean_kernel<false, NW, true>), grid, dim3(NW * 64), lds, stream, q);
  } else {
    if (p.causal) hipLaunchKernelGGL((td_attn_fwd_d128_lean_kernel<true, NW>), grid, dim3(NW * 64), lds, stream, q);
    else hipLaunchKernelGGL((td_attn_fwd_d128_lean_kernel<false, NW>), grid, dim3(NW * 64), lds, stream, q);
  }
  TD_CHECK_LAUNCH();
  return 0;
}
