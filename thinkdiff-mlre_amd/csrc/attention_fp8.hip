// Joint (unmasked) attention for gfx950 with BOTH products on the block-scaled 8-bit matrix instruction
// (v_mfma_scale_f32_32x32x64_f8f6f4, OCP e4m3 operands, 2x the bf16 MFMA rate): the attention of the FLUX engine's 8-bit modes
// (td_flux_set_attention(TD_ATTENTION_FP8)).  bf16 q / k / v in, bf16 (or history-scaled int8) out; head_dim = 128.
//
// Two kernels:
//  1. td_attn_fp8_pack_kernel: one pass over q | k | v that writes what the attention kernel's LDS wants, so that its tile loads
//     are linear 8 KiB copies:
//       q8  [Sq][H][128]            e4m3, q x (scale log2 e) under a power-of-two (E8M0) scale per (token, head)
//       k8  [H][tile][64 keys][128] e4m3 under an E8M0 scale per (key, head); 16-byte chunk c of key row r sits at chunk
//                                   c ^ ((r >> 1) & 7) (conflict-free for the ds_read_b128 lane groups on 128-byte rows)
//       v8t [H][tile][128 d][64]    e4m3 V^T under one E8M0 scale per (tile, head); the 64 keys of a tile are stored in the order the
//                                   S^T accumulators hand their probabilities to the P.V product (below), 16-byte chunk c of row d
//                                   at chunk c ^ ((d >> 2) & 3)
//     The scales ride on the MFMA's scale operands (one byte per lane, the same for both lane halves of a row -- the only form
//     whose block <-> lane map tools/probes/mfma_fp8_probe.hip pins), so no score or output is ever rescaled on the VALU.
//  2. td_attn_fwd_d128_fp8_kernel: the stream-K skeleton of attention_bf16.hip (persistent workgroups over equal (item, KV tile)
//     ranges, wait-free hand-off of split items) around an 8-bit tile body:
//       S^T - ref = K8 . Q8^T + C     2 x 2 MFMAs (K = 64 each); the row's reference point enters as the C operand (a resident
//                                     16-register tuple of -ref), scores arrive in the exp2 domain
//       P -> e4m3 byte                SHIPPED FORM ("LIN", the only one the engine launches): no exponential -- the byte is rne(8 (s - ref) + 56)
//                                     clamped to [0, 126], i.e. 2^floor(x) (1 + frac x) with 3 mantissa bits in place of 2^x (an e4m3 byte read
//                                     as an integer is a piecewise-linear log2 scale): up to 6.1 % above exp2 per probability, mean 4.3 %, which
//                                     the row sum -- taken over the same bytes -- cancels; 1.9 % rms remains.  The exp2 + v_cvt_pk_fp8_f32 form is
//                                     kept for A/B only (variant bit 0x2000 of the stand-alone entry; td_flux strips it).
//                                     The 32 bytes of a lane, converted in place, ARE the B operand of one K = 64 MFMA: byte
//                                     p = 16 kb + reg of lane half h is key 32 kb + (reg & 3) + 8 (reg >> 2) + 4 h, which is the key
//                                     order the pack kernel gave v8t's rows
//       O^T += V8^T . P^T             4 MFMAs, + 1 whose A operand is all ones = the row sums over the rounded probabilities
//     9 MFMAs of 64 cycles per 64-key tile and wave against 38 of 32 in the bf16 kernel; 16 ds_read_b128 against 16 + 32
//     transposed reads; 32 KiB of LDS.  The reference point is an INTEGER (ceil of a running maximum - 5) kept so that a row's
//     largest probability lies in (2^4, 2^8.75): e4m3 then resolves probabilities down to 2^-16 of the row maximum, and -- every
//     reference being a whole power of two away from any other -- a probability is rounded to the same 3 mantissa bits whatever the
//     tiling and the order of arrival were (the oracle's restatement uses ceil(row maximum) - 5 and agrees to fp32 rounding).
// Numerics: tests/test_attention_fp8_gpu.py compares the LIN kernel with oracle/flux_ref.py's restatement in ITS "linear" mode
// (FP8_ATTENTION_PROB = "linear", bytes equal up to fp32 rounding of the scores) and the A/B exp form with the "exp2" mode; the engine's
// 8-bit policies are graded end to end on the 28-step fixtures (plain and heavy-tailed checkpoint, tests/test_flux_full_depth_gpu.py).
#include "attention_common.h"
#include "qk_rope_math.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;   // one 8-bit MFMA operand: 32 bytes per lane

constexpr int TILE8 = KV_TILE * D;     // 8 KiB per K8 or V8^T tile
// Where a new reference point puts the row maximum: in (2^4, 2^5], i.e. 3.75 ... 4.75 octaves below the limit and 11 above the smallest byte.  The
// choice trades rescales against tail resolution (round 3, in the denoise loop, two alternated runs each): headroom 7 (maximum in (2^6, 2^7], the
// first form) 285 ms of attention per image, 5: 274 ms (+1.1 % images/s), 3: 263 ms (+3.2 %) -- every move of the reference costs ~80 VALU
// instructions per wave (O, row sums and scores rescaled) and on the engine's activations it moved in about every second tile.  Against the
// restatement at headroom 7 on Gaussian operands: 5 differs by 2.2e-4 (worst row 7e-3: nothing beside e4m3's own 5.6e-2), 3 by 4.0e-3 with a worst
// row of 9e-2 (rows whose maximum sits 9+ octaves above the bulk lose the bulk to the subnormal bytes); the 28-step pixel RMSE is 6.35e-3 for all three.
constexpr float REF_HEADROOM = 5.0f;
constexpr float REF_LIMIT = 8.75f;     // ... and moves again before a probability passes 2^8.75 = 430 (e4m3 tops out at 448)

struct F8Layout {
  size_t q8, qs, k8, ks, v8, vs, total;
};
F8Layout f8_layout(int Sq, int Skv, int H) {
  const size_t nt = (size_t)(Skv + KV_TILE - 1) / KV_TILE;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  F8Layout l;
  size_t o = 0;
  l.q8 = o; o += up((size_t)Sq * H * D);
  l.qs = o; o += up((size_t)Sq * H);
  l.k8 = o; o += up((size_t)H * nt * TILE8);
  l.ks = o; o += up((size_t)H * nt * KV_TILE);
  l.v8 = o; o += up((size_t)H * nt * TILE8);
  l.vs = o; o += up((size_t)H * nt);
  l.total = o;
  return l;
}

// The E8M0 byte b (value 2^(b - 127)) with amax / 2^(b - 127) <= 448, the smallest such, kept inside [2^-40, 2^40].
__device__ __forceinline__ unsigned e8m0_for(float amax) {
  const unsigned u = as_u32(amax * (1.0f / 448.0f));
  const unsigned b = ((u >> 23) & 0xffu) + ((u & 0x7fffffu) ? 1u : 0u);
  return min(max(b, 87u), 167u);
}
__device__ __forceinline__ float e8m0_inv(unsigned b) { return as_f32((254u - b) << 23); }   // 2^(127 - b)

// 32 floats -> 32 e4m3 bytes (8 words), x * mul each
__device__ __forceinline__ void to_e4m3_32(const float (&x)[32], const float mul, unsigned (&w)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i] * mul, x[4 * i + 1] * mul, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i + 2] * mul, x[4 * i + 3] * mul, v, true);
    w[i] = (unsigned)v;
  }
}

// A thread's 32 bf16 of a head row: fetched as four 16-byte loads (raw), unpacked where they are used -- the pack kernel issues the q, k and v
// fetches of a thread together, ahead of the first use, so the three sections do not each pay an HBM round trip of their own
__device__ __forceinline__ void fetch_row32(const bf16_t* p, bool live, u32x4_t (&raw)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    raw[i] = u32x4_t{0u, 0u, 0u, 0u};
    if (live) raw[i] = *(const u32x4_t*)(p + 8 * i);
  }
}
__device__ __forceinline__ void unpack_row32(const u32x4_t (&raw)[4], float (&x)[32]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[8 * i + 2 * j] = bf_lo(raw[i][j]); x[8 * i + 2 * j + 1] = bf_hi(raw[i][j]); }
}

// QK-RMSNorm + rotary embedding of td_qk_norm_rope_kernel on this thread's 32 elements of a head row (quarter qt of the row; the four
// threads of a row are lanes 4r .. 4r+3): group g of 8 elements is what lane 4 qt + g of that kernel's 16-lane row owns, and the sum of
// squares is added in its order -- the xor-butterfly 8, 4, 2, 1 over the 16 partial sums (8 and 4 cross threads: xor 2 and 1 here; 2
// and 1 are this thread's own groups) -- so both kernels round the same numbers.  Output: the bf16-rounded values, as floats.
__device__ __forceinline__ void norm_rope_32(float (&x)[32], const bf16_t* w, const int qt, const float eps, const float (&cs)[32], const float (&sn)[32], const float premul) {
  float g[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 8; ++i) g[a][i] = x[8 * a + i];
  if (w) {
    float sq[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) sq[a] = qk_sumsq8(g[a]);
#pragma unroll
    for (int a = 0; a < 4; ++a) sq[a] += __shfl_xor(sq[a], 2);      // lanes l, l ^ 8
#pragma unroll
    for (int a = 0; a < 4; ++a) sq[a] += __shfl_xor(sq[a], 1);      // ... ^ 4
    const float s2a = sq[0] + sq[2], s2b = sq[1] + sq[3];            // ... ^ 2
    const float rstd = qk_rstd(s2a + s2b, eps);                      // ... ^ 1
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float wv[8];
      const u32x4_t wr = *(const u32x4_t*)(w + qt * 32 + 8 * a);
#pragma unroll
      for (int j = 0; j < 4; ++j) { wv[2 * j] = bf_lo(wr[j]); wv[2 * j + 1] = bf_hi(wr[j]); }
      qk_norm8(g[a], rstd, wv);
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float c8[8], s8[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { c8[i] = cs[8 * a + i]; s8[i] = sn[8 * a + i]; }
    qk_rope_pairs8(g[a], c8, s8, y);
    if (premul != 1.0f) {
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] *= premul;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) x[8 * a + i] = rbf(y[i]);
  }
}

}  // namespace

// One workgroup per (64-token tile, head): thread (row = tid / 4, quarter = tid % 4) owns 32 of a token's 128 head dims.
// `probe` (TD_PACK_PROBE, timing experiments only -- the attention then runs on stale operands): bit 0 skips the v section, bit 1 k, bit 2 q.
__global__ __launch_bounds__(256) void td_attn_fp8_pack_kernel(const TdAttnParams p, char* __restrict__ ws, const F8Layout lay, const float qmul, const int nt, const int probe) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) uint8_t vimg[TILE8];
  __shared__ float wmax[4];
  const int tid = threadIdx.x, t = blockIdx.x, head = blockIdx.y;
  const int row = tid >> 2, qt = tid & 3;
  const int tok = t * KV_TILE + row;
  const int H = p.Hq;
  float x[32];
  unsigned w[8];
  // fused QK-RMSNorm + RoPE (p.rope_cos): this token's table row, shared by its q and k (Sq == Skv)
  const bool rope = p.rope_cos != nullptr;
  float cs[32], sn[32];
  if (rope && tok < p.Sq) {
    const float* cr = p.rope_cos + (size_t)tok * D + qt * 32;
    const float* sr = p.rope_sin + (size_t)tok * D + qt * 32;
#pragma unroll
    for (int i = 0; i < 32; i += 4) {
      const f32x4_t c4 = *(const f32x4_t*)(cr + i);
      const f32x4_t s4 = *(const f32x4_t*)(sr + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) { cs[i + j] = c4[j]; sn[i + j] = s4[j]; }
    }
  }
  const bool partB = tok >= p.rope_split;
  const bool live = tok < p.Skv;
  // all three rows of this thread are requested here (probe builds skip a section's fetch with the section)
  u32x4_t rq[4], rk[4], rv[4];
  fetch_row32(p.Q + (size_t)min(tok, p.Sq - 1) * p.ldq + head * D + qt * 32, tok < p.Sq && !(probe & 4), rq);
  fetch_row32(p.K + (size_t)(live ? tok : 0) * p.ldkv + head * D + qt * 32, live && t < nt && !(probe & 2), rk);
  fetch_row32(p.V + (size_t)(live ? tok : 0) * p.ldkv + head * D + qt * 32, live && t < nt && !(probe & 1), rv);

  if (tok < p.Sq && !(probe & 4)) {      // ---- q: row-major, scale per (token, head)
    unpack_row32(rq, x);
    if (rope) norm_rope_32(x, partB ? p.rope_wqB : p.rope_wqA, qt, p.rope_eps, cs, sn, p.rope_q_premul);
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { x[i] *= qmul; am = fmaxf(am, fabsf(x[i])); }
    am = fmaxf(am, __shfl_xor(am, 1));
    am = fmaxf(am, __shfl_xor(am, 2));
    const unsigned sb = e8m0_for(am);
    to_e4m3_32(x, e8m0_inv(sb), w);
    uint8_t* dst = (uint8_t*)ws + lay.q8 + ((size_t)tok * H + head) * D + qt * 32;
    *(u32x4_t*)dst = u32x4_t{w[0], w[1], w[2], w[3]};
    *(u32x4_t*)(dst + 16) = u32x4_t{w[4], w[5], w[6], w[7]};
    if (qt == 0) {
      ((uint8_t*)ws + lay.qs)[(size_t)tok * H + head] = (uint8_t)sb;
      // this launch's slice of the next step's reference points starts "far below any reference" (the attention kernel max-accumulates into it);
      // cleared here, by the pass that runs in front of it on the same stream, instead of a 24 MB memset over all blocks per denoise step
      if (p.ref_out) p.ref_out[(size_t)head * p.Sq + tok] = (int)0x80808080;
    }
  }
  if (t >= nt) return;   // (Sq > Skv: the remaining tiles carry queries only)
  const size_t tile = (size_t)head * nt + t;
  if (!(probe & 2)) {    // ---- k: the swizzled LDS image of the tile, scale per (key, head); rows past Skv are zero
    unpack_row32(rk, x);
    if (rope && live) norm_rope_32(x, partB ? p.rope_wkB : p.rope_wkA, qt, p.rope_eps, cs, sn, 1.0f);      // (live is uniform over a row's four threads)
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) am = fmaxf(am, fabsf(x[i]));
    am = fmaxf(am, __shfl_xor(am, 1));
    am = fmaxf(am, __shfl_xor(am, 2));
    const unsigned sb = e8m0_for(am);
    to_e4m3_32(x, e8m0_inv(sb), w);
    uint8_t* dst = (uint8_t*)ws + lay.k8 + tile * TILE8 + row * D;
    const int sw = (row >> 1) & 7;
    *(u32x4_t*)(dst + (((2 * qt) ^ sw) << 4)) = u32x4_t{w[0], w[1], w[2], w[3]};
    *(u32x4_t*)(dst + (((2 * qt + 1) ^ sw) << 4)) = u32x4_t{w[4], w[5], w[6], w[7]};
    if (qt == 0) ((uint8_t*)ws + lay.ks)[tile * KV_TILE + (row & 31) * 2 + (row >> 5)] = (uint8_t)sb;   // lane l31 reads its two k-blocks' bytes as one u16
  }
  if (!(probe & 1)) {    // ---- v: transposed through LDS, one scale per (tile, head), keys in the accumulator's order
    unpack_row32(rv, x);
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) am = fmaxf(am, fabsf(x[i]));
    am = wave_max(am);
    if ((tid & 63) == 0) wmax[tid >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const unsigned sb = e8m0_for(am);
    to_e4m3_32(x, e8m0_inv(sb), w);
    // key `row` of the tile -> byte position inside a V^T row: lane half h = bit 2, register reg = (key & 3) + 4 (key32 >> 3)
    const int k32 = row & 31;
    const int pos = 32 * ((k32 >> 2) & 1) + 16 * (row >> 5) + (k32 & 3) + 4 * (k32 >> 3);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int d = qt * 32 + i;
      vimg[d * KV_TILE + ((((pos >> 4) ^ ((d >> 2) & 3))) << 4) + (pos & 15)] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    }
    __syncthreads();
    uint8_t* dst = (uint8_t*)ws + lay.v8 + tile * TILE8 + tid * 32;
    *(u32x4_t*)dst = *(const u32x4_t*)(vimg + tid * 32);
    *(u32x4_t*)(dst + 16) = *(const u32x4_t*)(vimg + tid * 32 + 16);
    if (tid == 0) ((uint8_t*)ws + lay.vs)[tile] = (uint8_t)sb;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// Tile pipeline: a ring of four K8 | V8^T slots.  At the top of iteration t a wave waits (counted vmcnt) until ITS pieces of tile
// t+1 have landed, the barrier makes tiles <= t+1 visible to all and frees slot (t+3) % 4 (last read in iteration t-1), whose
// refill (two tiles ahead of its first reader) is issued at once.  The K fragments of tile t are already in registers -- read at
// the end of iteration t-1 under its P.V MFMAs -- so the score MFMAs start right after the barrier; the V^T fragments are read
// while they run.  No LDS or HBM latency is exposed inside an item.
//
// LIN (shipped): the probability's e4m3 byte is made by an INTEGER conversion, with no exponential at all.  An e4m3 byte read as an
// integer b is a piecewise-linear log2 scale -- b = 8 (E + M/8) stands for 2^(E - 7) (1 + M/8) -- so with the scores arriving as
// y = 8 (s - ref) + 56 (q carries the factor 8 in its power-of-two scale, the C operand 56 - 8 ref) the byte is simply
// rne(y), one v_cvt_pk_u8_f32 per score (saturating at 0: masked keys and everything below 2^-7 of the reference vanish; y <= 126 =
// the byte of 448 by the reference rule).  That is 2^floor(x) (1 + frac(x)) in place of 2^x: at most 6.1 % above it (mean 4.3 %,
// which the row sum -- taken over the same bytes -- cancels; 1.9 % rms remains, less than e4m3's own rounding of a probability).
// It replaces 32 v_exp_f32 (8 issue cycles each) + 16 v_cvt_pk_fp8_f32 per tile and wave by 32 4-cycle conversions: the exp form
// of this kernel is VALU-bound (rocprofv3: MFMA 35 %, VALU 61 % of the cycles, not overlapping), this one is not.
// PROBE (-DTD_ATTN8_PROBE builds only; results are WRONG, the launch duration is the measurement): 1 no row-maximum chain, 2 no reference logic at
// all, 3 probabilities not converted (the P.V MFMAs no longer wait for the scores), 4 no score MFMAs, 5 no P.V MFMAs, 6 no tile barrier.
template <int NWAVES, bool XCD_REMAP, bool LIN, int PROBE = 0>
__global__ __launch_bounds__(NWAVES * 64, 2) void td_attn_fwd_d128_fp8_kernel(const TdAttnParams p, const char* __restrict__ pk, const F8Layout lay,
                                                                               char* __restrict__ ws, const int n_qblk, const int nt) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NSLOT = 4;
  constexpr int PIECES = 8 / NWAVES;            // 1 KiB LDS-DMA pieces per wave, tile and operand
  constexpr int VMOPS = 2 * PIECES + 2;         // vector-memory operations of one stage() per wave
  static_assert(NWAVES == 8 || NWAVES == 4, "8 KiB tiles in 1 KiB pieces");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [K slot 0..3 | V slot 0..3 | ticket word]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5;
  const int l31 = lane & 31;
  unsigned* const cnt = (unsigned*)ws + 16;
  TD_LDS unsigned* const ticket_lds = (TD_LDS unsigned*)(smem + 2 * NSLOT * TILE8);

  const int G = gridDim.x;
  int r = blockIdx.x;
  if constexpr (XCD_REMAP) {
    const int q8 = G >> 3, r8 = G & 7, xcd = r & 7;
    r = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (r >> 3);
  }
  const long long total = (long long)n_qblk * p.Hq * nt;
  long long it = total * r / G;
  const long long it_end = total * (r + 1) / G;
  const int Skv = p.Skv, H = p.Hq;

  // ---- resident per-lane LDS byte addresses of slot 0 (slot, k-block and d-block parts are instruction immediates) ----
  const unsigned lds0 = (unsigned)(uintptr_t)(TD_LDS char*)smem;
  unsigned ka[4];      // K row read [ks * 2 + j]: key row l31, 16-byte chunk (4 ks + 2 h5 + j) ^ ((row >> 1) & 7)
#pragma unroll
  for (int e = 0; e < 4; ++e) ka[e] = lds0 + l31 * D + ((((e >> 1) * 4 + 2 * h5 + (e & 1)) ^ ((l31 >> 1) & 7)) << 4);
  unsigned va[2];      // V^T row read [j]: row d = l31, chunk (2 h5 + j) ^ ((d >> 2) & 3)
#pragma unroll
  for (int j = 0; j < 2; ++j) va[j] = lds0 + NSLOT * TILE8 + l31 * KV_TILE + (((2 * h5 + j) ^ ((l31 >> 2) & 3)) << 4);
  const i32x8_t ones = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};   // e4m3 1.0

  const __amdgpu_buffer_rsrc_t rsQ8 = __builtin_amdgcn_make_buffer_rsrc((void*)(pk + lay.q8), 0, (unsigned)((size_t)p.Sq * H * D), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsQs = __builtin_amdgcn_make_buffer_rsrc((void*)(pk + lay.qs), 0, (unsigned)((size_t)p.Sq * H), 0x00020000);

  while (it < it_end) {
    const int item = (int)(it / nt);
    const int kb = (int)(it - (long long)item * nt);
    const int ke = (int)min((long long)nt, kb + (it_end - it));
    __builtin_assume(ke > kb);
    it += ke - kb;
    const int qblk = item % n_qblk;
    const int head = item / n_qblk;
    const int q0 = qblk * (NWAVES * Q_WAVE) + wid * Q_WAVE;

    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(pk + lay.k8 + (size_t)head * nt * TILE8), 0, (unsigned)nt * TILE8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(pk + lay.v8 + (size_t)head * nt * TILE8), 0, (unsigned)nt * TILE8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsKs = __builtin_amdgcn_make_buffer_rsrc((void*)(pk + lay.ks + (size_t)head * nt * KV_TILE), 0, (unsigned)nt * KV_TILE, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsVs = __builtin_amdgcn_make_buffer_rsrc((void*)(pk + lay.vs + (size_t)head * nt), 0, (unsigned)nt, 0x00020000);
    unsigned ksc_r[NSLOT], vsc_r[NSLOT];      // scale bytes of the tiles in the ring
    // Every iteration stages exactly one tile (indices past the end re-load the last one into a slot nobody reads): the number of
    // vector-memory operations in flight is then the same at every barrier, and the staging code has no branch -- it sits
    // inside the MFMA streams (an LDS-DMA piece costs ~60 issue cycles, invisible under a 64-cycle MFMA, and ~600 per tile when all
    // eight waves issue theirs together right after the barrier: rocprofv3 + ablations), see stage_part below.
    auto stage = [&](auto slot_tag, int t_raw) {
      constexpr int SL = decltype(slot_tag)::value;
      const unsigned t = (unsigned)min(t_raw, nt - 1);
#pragma unroll
      for (int pc = 0; pc < PIECES; ++pc) {
        const unsigned piece = (unsigned)(wid * PIECES + pc) * 1024u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (TD_LDS void*)(smem + SL * TILE8 + piece), 16, piece + (unsigned)lane * 16u + t * TILE8, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (TD_LDS void*)(smem + (NSLOT + SL) * TILE8 + piece), 16, piece + (unsigned)lane * 16u + t * TILE8, 0, 0, 0);
      }
      ksc_r[SL] = __builtin_amdgcn_raw_buffer_load_b16(rsKs, t * KV_TILE + l31 * 2, 0, 0);
      vsc_r[SL] = __builtin_amdgcn_raw_buffer_load_b8(rsVs, t, 0, 0);
    };
    // In the tile loop a stage goes out in two halves: the K piece (+ its scale bytes) behind the four score MFMAs, where the wave would only be waiting
    // for their results, the V piece between the P.V MFMAs.  Measured in the denoise loop, alternated twice: whole stage between the P.V MFMAs 272.7 / 273.0 ms
    // of attention per image, this split 267.0 / 266.2, whole stage behind the score MFMAs 270.3 / 269.5 (profiles/r3c_attention_fp8_stage_split_ab.log).
    auto stage_part = [&](auto slot_tag, int t_raw, bool vpart) {
      constexpr int SL = decltype(slot_tag)::value;
      const unsigned t = (unsigned)min(t_raw, nt - 1);
#pragma unroll
      for (int pc = 0; pc < PIECES; ++pc) {
        const unsigned piece = (unsigned)(wid * PIECES + pc) * 1024u;
        if (!vpart) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (TD_LDS void*)(smem + SL * TILE8 + piece), 16, piece + (unsigned)lane * 16u + t * TILE8, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (TD_LDS void*)(smem + (NSLOT + SL) * TILE8 + piece), 16, piece + (unsigned)lane * 16u + t * TILE8, 0, 0, 0);
      }
      if (!vpart) ksc_r[SL] = __builtin_amdgcn_raw_buffer_load_b16(rsKs, t * KV_TILE + l31 * 2, 0, 0);
      else vsc_r[SL] = __builtin_amdgcn_raw_buffer_load_b8(rsVs, t, 0, 0);
    };
    auto kread = [&](auto slot_tag, int kbk, int ks) {
      constexpr unsigned PO = decltype(slot_tag)::value * TILE8;
      const u32x4_t a = *(const TD_LDS u32x4_t*)(uintptr_t)(ka[2 * ks] + PO + kbk * 32 * D);
      const u32x4_t b = *(const TD_LDS u32x4_t*)(uintptr_t)(ka[2 * ks + 1] + PO + kbk * 32 * D);
      return i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the previous part's surplus stages and output stores are done ...
    __builtin_amdgcn_s_barrier();                                     // ... and so are everyone's reads of the ring
    stage(S0{}, kb);
    i32x8_t qf[2];
    unsigned qsc;
    {
      const unsigned qrow = (unsigned)min(q0 + l31, p.Sq - 1);
      const unsigned qoff = (qrow * (unsigned)H + (unsigned)head) * D + 32 * h5;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rsQ8, qoff + ks * 64, 0, 0);
        const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(rsQ8, qoff + ks * 64 + 16, 0, 0);
        qf[ks] = i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
      }
      qsc = __builtin_amdgcn_raw_buffer_load_b8(rsQs, qrow * (unsigned)H + (unsigned)head, 0, 0);
    }
    stage(S1{}, kb + 1);
    stage(S2{}, kb + 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * VMOPS) : "memory");      // tile kb (and q) landed: everything but the two younger stages
    __builtin_amdgcn_s_barrier();
    i32x8_t kf[4];      // K fragments of the coming tile [kb * 2 + ks]
    kf[0] = kread(S0{}, 0, 0); kf[2] = kread(S0{}, 1, 0); kf[1] = kread(S0{}, 0, 1); kf[3] = kread(S0{}, 1, 1);

    f32x16_t o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) o[db][rr] = 0.f;
    f32x16_t lacc, negm;      // row sums (every register the whole sum); -reference, the C operand of the score chain
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) { lacc[rr] = 0.f; negm[rr] = LIN ? 56.f : 0.f; }
    float m_run = 0.f;      // the reference point, log2 units
    // History reference (p.ref_in): the row starts where the previous denoise step's largest score put it.  `need_first`: a row without one
    // (or no history at all) takes the first tile's maximum, as before; a row whose history turns out far too high for its first tile does too.
    float smax_run = -INFINITY;      // the row's largest score so far, absolute log2 units (for p.ref_out)
    bool need_first = true;
    if (p.ref_in) {
      const int href = p.ref_in[(size_t)head * p.Sq + min(q0 + l31, p.Sq - 1)];
      need_first = __any(href < -(1 << 20));
      if (!need_first) {
        m_run = (float)href;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) negm[rr] = LIN ? 56.f - 8.f * m_run : -m_run;
      }
    }

    auto tile = [&](const int t, auto slot_tag) {
      constexpr int SLOT = decltype(slot_tag)::value;
      constexpr unsigned PO = SLOT * TILE8;
      using NEXT = std::integral_constant<int, (SLOT + 1) % NSLOT>;
      using FREE = std::integral_constant<int, (SLOT + 3) % NSLOT>;
      // my pieces of tile t+1 landed (tile t+2 may still be in flight); after the barrier everyone's are visible and slot FREE
      // (tile t-1) has no reader left
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMOPS) : "memory");
      if constexpr (PROBE != 6) __builtin_amdgcn_s_barrier();
      const int ksc = (int)ksc_r[SLOT], vsc = (int)vsc_r[SLOT];

      // ---- S^T - ref = K8 . Q8^T - ref ------------------------------------------------------------------------------------
      f32x16_t st[2];
      i32x8_t vf[4];
      auto vread = [&](int db) {
        const u32x4_t a = *(const TD_LDS u32x4_t*)(uintptr_t)(va[0] + PO + db * 32 * KV_TILE);
        const u32x4_t b = *(const TD_LDS u32x4_t*)(uintptr_t)(va[1] + PO + db * 32 * KV_TILE);
        return i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
      };
      const int key0 = t * KV_TILE;
      auto mask_tail = [&]() {        // last tile: keys >= Skv are masked
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            const int key = key0 + kk * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h5;
            if (key >= Skv) st[kk][rr] = -INFINITY;
          }
      };
      auto convert = [&](int kk, i32x8_t& pf) {      // P of key half kk as e4m3 bytes, in the byte order of the P.V product's B operand
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          unsigned wv = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) wv = __builtin_amdgcn_cvt_pk_u8_f32(st[kk][4 * g + b], b, wv);
          pf[kk * 4 + g] = (int)wv;
        }
      };
      i32x8_t pf;
      const bool first = PROBE == 2 ? false : (t == kb && need_first);
      bool fix_low = false;      // (history) rows whose first tile sits in the subnormal bytes of their inherited reference start from the tile after all
      constexpr float LIMIT = LIN ? 8.f * REF_LIMIT + 56.f : REF_LIMIT;      // (LIN: 126, the byte of 448)
      constexpr float LOW = LIN ? 8.f : -6.f;                                  // (LIN: the first normal byte, 2^-6)
      // the reference point moves: first tile of a part, or a probability about to leave e4m3's range (mx: the row maximum, reference-relative)
      auto move_reference = [&](const float mx) {
        const float xm = LIN ? (mx - 56.f) * 0.125f : mx;                                   // the row maximum in log2 units above the reference
        float d = (first || mx > LIMIT || (fix_low && mx < LOW)) ? __builtin_ceilf(xm) - REF_HEADROOM : 0.f;      // the new reference relative to the old one (per row): an integer
        if (!(d > -INFINITY)) d = 0.f;
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
          for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) o[db][rr] *= alpha;
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) lacc[rr] *= alpha;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) st[kk][rr] -= LIN ? 8.f * d : d;
        m_run += d;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) negm[rr] = LIN ? 56.f - 8.f * m_run : -m_run;
      };
      // four MFMAs, the V^T fragments of this tile read in their shadow (into the registers the K fragments leave)
      if constexpr (PROBE == 4) {
        st[0] = negm; st[1] = negm;
        vf[0] = vread(0); vf[1] = vread(1); vf[2] = vread(2); vf[3] = vread(3);
      } else {
      st[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf[0], qf[0], negm, 0, 0, 0, ksc, 0, (int)qsc);
      vf[0] = vread(0);
      st[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf[2], qf[0], negm, 0, 0, 1, ksc, 0, (int)qsc);
      vf[1] = vread(1);
      st[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf[1], qf[1], st[0], 0, 0, 0, ksc, 0, (int)qsc);
      vf[2] = vread(2);
      st[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf[3], qf[1], st[1], 0, 0, 1, ksc, 0, (int)qsc);
      vf[3] = vread(3);
      }
      stage_part(FREE{}, t + 3, false);      // (behind the four score MFMAs in program order: the wave would only be waiting for them)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }

      if (key0 + KV_TILE > Skv) mask_tail();

      // ---- the reference point: first tile of a part, or a probability about to leave e4m3's range ---------------------------
      // The first read of the fresh accumulators is one the compiler can see: it places the wait states a VALU read of a 16-pass
      // MFMA result needs.  The inline-asm v_max3 alone is invisible to its hazard recogniser and read registers the matrix pipe
      // had not written yet (a row maximum that missed elements -> a probability past 448 -> NaN, about one row in 600).
      float mx = fmaxf(st[0][15], st[1][15]);
      if constexpr (PROBE != 1 && PROBE != 2) {
#pragma unroll
        for (int rr = 0; rr < 15; ++rr) mx = max3(mx, st[0][rr], st[1][rr]);
        mx = half_swap_max(mx);
      }
      if constexpr (PROBE == 2) mx = 0.f;
      if (p.ref_out) smax_run = fmaxf(smax_run, LIN ? __builtin_fmaf(mx - 56.f, 0.125f, m_run) : mx + m_run);
      // A row trusts its history even when its first tile is far below it (the peak comes later: that is what the history knows) -- unless the tile
      // sits in the subnormal bytes (11+ octaves under where the row's maximum was last step): such a row starts from this tile like a row without
      // history, so that a stale reference can never make a whole row underflow (some tile always contributes normal bytes).
      if (t == kb && !need_first) fix_low = __any(mx < LOW);
      if (first || fix_low || __any(mx > LIMIT)) move_reference(mx);

      // ---- P = exp2(.) as e4m3, in the byte order of the P.V product's B operand -------------------------------------------------
      if constexpr (LIN) {
        convert(0, pf);
        convert(1, pf);
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            int wv = 0;
            wv = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_exp2f(st[kk][4 * g]), __builtin_amdgcn_exp2f(st[kk][4 * g + 1]), wv, false);
            wv = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_exp2f(st[kk][4 * g + 2]), __builtin_amdgcn_exp2f(st[kk][4 * g + 3]), wv, true);
            pf[kk * 4 + g] = wv;
          }
      }

      if constexpr (PROBE == 3) pf = qf[0];
      // ---- O^T += V8^T . P^T, row sums as one more row-block of ones; the next tile's K fragments under these MFMAs ---------
      lacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, pf, lacc, 0, 0, 0, 127, 0, 127);
      stage_part(FREE{}, t + 3, true);
      if constexpr (PROBE != 5) o[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf[0], pf, o[0], 0, 0, 0, vsc, 0, 127);
      kf[0] = kread(NEXT{}, 0, 0);
      if constexpr (PROBE != 5) o[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf[1], pf, o[1], 0, 0, 0, vsc, 0, 127);
      kf[2] = kread(NEXT{}, 1, 0);
      if constexpr (PROBE != 5) o[2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf[2], pf, o[2], 0, 0, 0, vsc, 0, 127);
      kf[1] = kread(NEXT{}, 0, 1);
      if constexpr (PROBE != 5) o[3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf[3], pf, o[3], 0, 0, 0, vsc, 0, 127);
      kf[3] = kread(NEXT{}, 1, 1);
      // MFMA | the stage's vector-memory operations, spread | MFMA | ... ; the next tile's K fragments under the last three
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, PIECES, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, PIECES, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    };
    for (int t = kb;;) {
      tile(t, S0{}); if (++t >= ke) break;
      tile(t, S1{}); if (++t >= ke) break;
      tile(t, S2{}); if (++t >= ke) break;
      tile(t, S3{}); if (++t >= ke) break;
    }
    if (p.ref_out && h5 == 0 && q0 + l31 < p.Sq && smax_run > -INFINITY)      // this part's share of the row's largest score, as the next step's starting reference
      __hip_atomic_fetch_max(p.ref_out + (size_t)head * p.Sq + q0 + l31, (int)__builtin_ceilf(smax_run) - (int)REF_HEADROOM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float l_run = lacc[0];
    // ---- hand-off of a split item: attention_bf16.hip's protocol, unchanged (state = O, reference, row sum) -----------------
    if (kb > 0 || ke < nt) {
      const int j = kb > 0 ? r : r + 1;
      const bool tail = kb > 0;
      const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + SK_HEADER_BYTES), 0, (unsigned)(2 * G) * SK_SLOT_FLOATS * 4u, 0x00020000);
      const unsigned lane_off = ((unsigned)wid * 17u * 64u + (unsigned)lane) * 16u;
      const unsigned mine = (unsigned)(tail ? j : G + j) * (SK_SLOT_FLOATS * 4u) + lane_off;
      const unsigned theirs = (unsigned)(tail ? G + j : j) * (SK_SLOT_FLOATS * 4u) + lane_off;
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
        for (int q4 = 0; q4 < 16; ++q4) {
          const u32x4_t v = {as_u32(o[q4 >> 2][4 * (q4 & 3)]), as_u32(o[q4 >> 2][4 * (q4 & 3) + 1]),
                             as_u32(o[q4 >> 2][4 * (q4 & 3) + 2]), as_u32(o[q4 >> 2][4 * (q4 & 3) + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, rsS, mine + q4 * 1024u, 0, 16);       // aux 16 = sc1
        }
        const u32x4_t ml = {as_u32(m_run), as_u32(l_run), 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(ml, rsS, mine + 16 * 1024u, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        other_ready = ask(true) == 1u;
        if (!other_ready) continue;
      }
      if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const u32x4_t ml = __builtin_amdgcn_raw_buffer_load_b128(rsS, theirs + 16 * 1024u, 0, 0);
      const float m2 = as_f32(ml[0]), l2 = as_f32(ml[1]);
      const float mm = fmaxf(m_run, m2);
      const float a1 = __builtin_amdgcn_exp2f(m_run - mm), a2 = __builtin_amdgcn_exp2f(m2 - mm);
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
      if (tid == 0) __hip_atomic_store(cnt + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    const float inv = 1.0f / l_run;
    const int q = q0 + l31;
    if (p.q8) {
      const int qr = min(q, p.Sq - 1);
      attn_store_rows_q8(o, inv, p.q8 + (size_t)qr * p.ldq8 + head * D, h5, p.q8_inv[qr], p.q8_amax + qr, q < p.Sq);
    } else {
      attn_store_rows(o, inv, p.O + (size_t)min(q, p.Sq - 1) * p.ldo + head * D, h5, (p.ldo & 7) == 0, q < p.Sq);
    }
  }
#endif
}

size_t td_attn_fp8_ws_bytes(int Sq, int Skv, int H) { return f8_layout(Sq, Skv, H).total; }

// p.f8_ws: td_attn_fp8_ws_bytes(Sq, Skv, Hq) bytes of scratch the two kernels share (contents undefined before and after);
// p.sk_ws: the stream-K hand-off workspace, as for td_attn_launch.
int td_attn_fp8_launch(const TdAttnParams& p, hipStream_t stream) {
  TD_CHECK_ARG(p.head_dim == D, "td_attention_fp8: head_dim=%d unsupported (only 128)", p.head_dim);
  TD_CHECK_ARG(p.Sq > 0 && p.Skv > 0 && p.Hq > 0 && p.batch == 1 && p.Hq == p.Hkv, "td_attention_fp8: one batch entry, Hq == Hkv, non-empty");
  TD_CHECK_ARG(!p.causal && !p.bias && !p.kv_lens && !p.seg_starts, "td_attention_fp8: joint (unmasked) attention only");
  TD_CHECK_ARG(p.ldq % 8 == 0 && p.ldkv % 8 == 0 && p.ldo % 4 == 0, "td_attention_fp8: row strides must be 16-byte multiples");
  TD_CHECK_ARG(((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V | (uintptr_t)p.O | (uintptr_t)p.f8_ws) % 16 == 0 && p.f8_ws, "td_attention_fp8: pointers must be 16-byte aligned, workspace present");
  if (p.rope_cos) TD_CHECK_ARG(p.rope_sin && p.Sq == p.Skv && ((uintptr_t)p.rope_cos | (uintptr_t)p.rope_sin | (uintptr_t)p.rope_wqA | (uintptr_t)p.rope_wkA | (uintptr_t)p.rope_wqB | (uintptr_t)p.rope_wkB) % 16 == 0,
                               "td_attention_fp8: fused QK-norm + RoPE needs both tables, Sq == Skv and 16-byte aligned tables / weights");
  if (p.q8) TD_CHECK_ARG(p.q8_inv && p.q8_amax && p.ldq8 % 8 == 0 && ((uintptr_t)p.q8) % 8 == 0, "td_attention_fp8: int8 output needs scales, maxima and 8-byte aligned rows");
  const int nt = (p.Skv + KV_TILE - 1) / KV_TILE;
  const F8Layout lay = f8_layout(p.Sq, p.Skv, p.Hq);
  TD_CHECK_ARG((size_t)p.Sq * p.Hq * D < (1ull << 32) && (size_t)nt * TILE8 < (1ull << 32), "td_attention_fp8: operand exceeds the 4 GiB buffer-descriptor range");
  // variant bit 0x1000 (A/B): 4-wave workgroups, two per CU (independent barriers, so the two waves of a SIMD drift apart) instead of one of 8
  const int NW = (p.variant & 0x1000) ? 4 : 8;
  const int n_qblk = (p.Sq + NW * Q_WAVE - 1) / (NW * Q_WAVE);
  const int n_items = n_qblk * p.Hq;
  int dev = 0;
  TD_CHECK_HIP(hipGetDevice(&dev));
  const int cus = td_attn_device_cus(dev);
  TD_CHECK_ARG(cus > 0 && 2 * cus < SK_MAX_RANGES && (long long)n_items * nt < (1ll << 31), "td_attention_fp8: device / problem outside the stream-K range");
  char* ws = (char*)p.sk_ws;
  if (!ws || NW == 4) {      // (the 4-wave A/B form has twice the ranges: a pooled workspace of its own)
    if (int rc = td_attn_pooled_workspace(dev, cus * (8 / NW), stream, &ws)) return rc;
  }
  const bool lin = !(p.variant & 0x2000);          // variant bit 0x2000 (A/B, tests): probabilities by exp2 + e4m3 conversion instead of the integer form
  const float qmul = (p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f) * (lin ? 8.0f : 1.0f);
  const int nt_pack = max(nt, (p.Sq + KV_TILE - 1) / KV_TILE);
  const char* pack_probe = getenv("TD_PACK_PROBE");
  hipLaunchKernelGGL(td_attn_fp8_pack_kernel, dim3(nt_pack, p.Hq), dim3(256), 0, stream, p, (char*)p.f8_ws, lay, qmul, nt, pack_probe ? atoi(pack_probe) : 0);
  TD_CHECK_LAUNCH();
  const int G = min(cus * (8 / NW), n_items);       // a range is never shorter than an item: every item is split over at most two workgroups
  constexpr int lds = 8 * TILE8 + 16;
  static std::atomic<unsigned long long> done[4] = {};
  auto go = [&](auto kernel, std::atomic<unsigned long long>& once, int threads) -> int {
    if (int e = td_attn_set_lds_attr((const void*)kernel, lds, once, dev)) return e;
    hipLaunchKernelGGL(kernel, dim3(G), dim3(threads), lds, stream, p, (const char*)p.f8_ws, lay, ws, n_qblk, nt);
    return 0;
  };
  int rc;
#ifdef TD_ATTN8_PROBE
  static std::atomic<unsigned long long> pdone[7] = {};
  switch ((p.variant >> 16) & 7) {
    case 1: return go(td_attn_fwd_d128_fp8_kernel<8, true, true, 1>, pdone[1], 512);
    case 2: return go(td_attn_fwd_d128_fp8_kernel<8, true, true, 2>, pdone[2], 512);
    case 3: return go(td_attn_fwd_d128_fp8_kernel<8, true, true, 3>, pdone[3], 512);
    case 4: return go(td_attn_fwd_d128_fp8_kernel<8, true, true, 4>, pdone[4], 512);
    case 5: return go(td_attn_fwd_d128_fp8_kernel<8, true, true, 5>, pdone[5], 512);
    case 6: return go(td_attn_fwd_d128_fp8_kernel<8, true, true, 6>, pdone[6], 512);
    default: break;
  }
#endif
  if (NW == 8) rc = lin ? go(td_attn_fwd_d128_fp8_kernel<8, true, true>, done[0], 512) : go(td_attn_fwd_d128_fp8_kernel<8, true, false>, done[1], 512);
  else rc = lin ? go(td_attn_fwd_d128_fp8_kernel<4, true, true>, done[2], 256) : go(td_attn_fwd_d128_fp8_kernel<4, true, false>, done[3], 256);
  if (rc) return rc;
  TD_CHECK_LAUNCH();
  return 0;
}
