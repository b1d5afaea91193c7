// bf16 "NT" GEMM for gfx950:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )   (fp32 accumulate)
//
// This is the one dense-contraction kernel of the FLUX / aligner / Qwen2-VL hot path
// (SURVEY.md 2.3 rows K1,K2,K4,K5,K6,K8,K12,K13,K14,K15,K19).  Both operands are K-contiguous
// (torch.nn.Linear weight layout), so the same fragment reader serves both sides.
//
// Structure (MI355X_MICROARCH / cdna_hip_programming guide 5):
//  * 8 waves (2 x 4), block tile BM x BN = (32*WM) x (64*WN), BK = 64, v_mfma_f32_16x16x32_bf16.
//  * Operand tiles go HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB = 8 rows x 128 B
//    per wave-instruction).  The LDS image is lane-linear, so the bank-conflict XOR swizzle
//    (16-B chunk ^= row&7) is applied to the per-lane SOURCE address and again on the ds_read.
//  * The buffer descriptors carry the true extents: rows >= M (or >= N) read as zero, which is
//    how ragged M (e.g. the 4289-token joint sequence) is handled without padding copies.
//  * MFMA operand roles are swapped (W fragment = "A" operand) and the W rows are staged in a
//    permuted order, so that every lane ends with 4*WN CONTIGUOUS output columns of one output
//    row: the epilogue (bias / activation / gate*x+residual) works on 16-B vectors and the
//    stores are full 128-B lines per row, with no LDS round trip.
//  * blockIdx -> tile mapping is XCD-aware (bijective remap + grouped M ordering) so the 32
//    tiles resident on one XCD share A/W panels through that XCD's L2.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

#include "td_common.h"
#include "td_kernels.h"

namespace {

[[maybe_unused]] constexpr int BK = 64;          // bf16 elements per K tile (= one 128-B LDS row)
constexpr int ROW_BYTES = 128;  // LDS row pitch

template <int ACT>
__device__ __forceinline__ float apply_act(float x) {
  if constexpr (ACT == TD_ACT_GELU_TANH) return gelu_tanh_f(x);
  else if constexpr (ACT == TD_ACT_GELU_ERF) return gelu_erf_f(x);
  else if constexpr (ACT == TD_ACT_SILU) return silu_f(x);
  else if constexpr (ACT == TD_ACT_QUICK_GELU) return quick_gelu_f(x);
  else return x;
}

// Epilogue.  A lane owns NV = 4*WN contiguous output columns [nbeg, nbeg+NV) of rows mbeg + 16*i.
// Rounding points follow the reference's bf16 torch pipeline: Linear output, activation, gate multiply and residual add
// each round to bf16.
// Every global access goes through a buffer descriptor that carries the true extent of its tensor: rows >= M, a null
// bias / residual and column chunks past N are range-checked by the hardware (loads return 0, stores are dropped), so the
// epilogue is straight-line code -- no per-access exec-mask branch, no 64-bit address arithmetic.  (The first form, with an
// `if (row ok && column ok)` around each access, compiled to ~20 k instructions with 24 spilled VGPRs in the 256x256 kernel;
// at one workgroup per CU the epilogue is dead time for the matrix pipe.)
// the problem a block works on (grouped launches carry two; see TdGemmParams)
struct ProbView {
  const bf16_t* bias; const bf16_t* gate; const bf16_t* res; bf16_t* C; int M;
  uint8_t* q8; const float* q8_inv; unsigned* q8_amax;      // int8 output form (TdGemmParams::q8), rows of THIS problem
  const bf16_t* q8_smooth;                                    // ... and its per-column smoothing factors (may be null)
};

constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;   // beyond any descriptor range (operands are < 4 GiB - 64 KiB, checked on the host)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, p ? bytes : 0u, 0x00020000);
}

// `act` and `mode` are block-uniform run-time values: ONE body, with scalar branches around the optional stages of a row.
// (One instantiation per activation, selected by a switch in front, made the compiler hoist the shared `acc + bias` of all
// WM rows above the switch: 128 extra live values and a spilling kernel.)
template <int WM, int WN, bool Q8 = false>   // mode 0: bias(+act); 1: bias, gate, (+res); 2: bias, res
__device__ __forceinline__ void epilogue(const TdGemmParams& pp, const ProbView& p, f32x4_t (&acc)[WN][WM], int mbeg, int nbeg, bool second,
                                         const int act, const int mode) {
  constexpr int NV = 4 * WN;
  constexpr int CH = (NV % 8 == 0) ? 8 : 4;          // columns per access: 16-byte accesses when the lane's span allows
  constexpr int NCH = NV / CH;
  bf16_t* Cout = second ? pp.C2 : p.C;
  const int ldo = second ? pp.ldc2 : pp.ldc;
  const int ncol = second ? nbeg - pp.n_split : nbeg;
  const int ncols_out = second ? pp.N - pp.n_split : (pp.C2 ? pp.n_split : pp.N);
  const bool use_res = p.res != nullptr;

  // Q8 (int8 kernels, 16 columns per lane): this output -- the activated one: the second of a split launch, else the only one -- leaves as
  // symmetric int8 under a per-row scale the CALLER fixed in advance (q8_inv[m] = 1 / scale: the engine takes it from the previous denoise
  // step), 16 bytes per lane and row, and the row maxima of what was produced go to q8_amax[m] (atomic max on the float bits: next step's
  // scales).  The separate per-token quantisation pass over this tensor then does not exist.
  const bool q8_here = Q8 && p.q8 != nullptr && (pp.C2 ? second : true);
  const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(q8_here ? p.q8 : nullptr, (unsigned)((long long)(p.M - 1) * pp.ldq8 + ncols_out));
  const __amdgpu_buffer_rsrc_t rsQi = make_rsrc(q8_here ? p.q8_inv : nullptr, (unsigned)p.M * 4u);
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(Cout, (unsigned)(((long long)(p.M - 1) * ldo + ncols_out) * 2));
  const __amdgpu_buffer_rsrc_t rsR = make_rsrc(p.res, (unsigned)(((long long)(p.M - 1) * pp.ldr + pp.N) * 2));
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.bias, (unsigned)pp.N * 2u);
  // (the int8 output form has no gate: its slot of per-column registers carries the smoothing factors instead)
  const bool q8_sm = q8_here && p.q8_smooth != nullptr;
  const __amdgpu_buffer_rsrc_t rsG = make_rsrc(q8_here ? p.q8_smooth : p.gate, (unsigned)pp.N * 2u);

  // per-column operands of the lane's NV columns (a null bias / gate reads as zeros; mode 1 always has a gate)
  float bias[NV], gate[NV];
  unsigned coff[NCH];            // byte offset of chunk c inside an output row
  bool cok[NCH];                 // chunk c lies inside N (N % 8 == 0: a chunk is entirely inside or entirely outside)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const bool inr = nbeg + c * CH + CH <= pp.N;
    cok[c] = inr;
    coff[c] = (unsigned)(ncol + c * CH) * 2u;
    if constexpr (CH == 8) {
      const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(nbeg + c * CH) * 2u, 0, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) { bias[c * 8 + 2 * k] = bf_lo(b[k]); bias[c * 8 + 2 * k + 1] = bf_hi(b[k]); }
      {
        const u32x4_t g = __builtin_amdgcn_raw_buffer_load_b128(rsG, (unsigned)(nbeg + c * CH) * 2u, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) { gate[c * 8 + 2 * k] = bf_lo(g[k]); gate[c * 8 + 2 * k + 1] = bf_hi(g[k]); }
      }
    } else {
      const u32x2_t b = __builtin_amdgcn_raw_buffer_load_b64(rsB, (unsigned)(nbeg + c * CH) * 2u, 0, 0);
      bias[c * 4] = bf_lo(b[0]); bias[c * 4 + 1] = bf_hi(b[0]); bias[c * 4 + 2] = bf_lo(b[1]); bias[c * 4 + 3] = bf_hi(b[1]);
      {
        const u32x2_t g = __builtin_amdgcn_raw_buffer_load_b64(rsG, (unsigned)(nbeg + c * CH) * 2u, 0, 0);
        gate[c * 4] = bf_lo(g[0]); gate[c * 4 + 1] = bf_hi(g[0]); gate[c * 4 + 2] = bf_lo(g[1]); gate[c * 4 + 3] = bf_hi(g[1]);
      }
    }
  }

  auto row = [&](const int i) {
    const int m = mbeg + i * 16;
    float v[NV];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[j][i][r];
    // Rounding points of the bf16 torch graph: the Linear output rounds before any following op, and so does every
    // following op; when nothing follows, the final pack is that rounding.
    auto round_all = [&]() {
#pragma unroll
      for (int c = 0; c < NV; c += 2) {
        const unsigned u = pack_bf2(v[c], v[c + 1]);
        v[c] = bf_lo(u);
        v[c + 1] = bf_hi(u);
      }
    };
#pragma unroll
    for (int c = 0; c < NV; ++c) v[c] += bias[c];
    if (act != TD_ACT_NONE) {
      round_all();
      switch (act) {
        case TD_ACT_GELU_TANH:
#pragma unroll
          for (int c = 0; c < NV; ++c) v[c] = apply_act<TD_ACT_GELU_TANH>(v[c]);
          break;
        case TD_ACT_GELU_ERF:
#pragma unroll
          for (int c = 0; c < NV; ++c) v[c] = apply_act<TD_ACT_GELU_ERF>(v[c]);
          break;
        case TD_ACT_SILU:
#pragma unroll
          for (int c = 0; c < NV; ++c) v[c] = apply_act<TD_ACT_SILU>(v[c]);
          break;
        default:
#pragma unroll
          for (int c = 0; c < NV; ++c) v[c] = apply_act<TD_ACT_QUICK_GELU>(v[c]);
          break;
      }
    }
    if constexpr (Q8 && NV == 16) {
      if (q8_here) {
        round_all();                     // the activation's output is a bf16 tensor in the reference graph
        if (q8_sm) {
#pragma unroll
          for (int c = 0; c < NV; ++c) v[c] *= gate[c];      // 1 / s of the consuming Linear's smoothing: a power of two, exact
        }
        float am = 0.f;
#pragma unroll
        for (int c = 0; c < NV; ++c) am = fmaxf(am, fabsf(v[c]));
        am = fmaxf(am, __shfl_xor(am, 16, 64));
        am = fmaxf(am, __shfl_xor(am, 32, 64));          // the row's maximum over this wave's 64 columns
        const float inv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsQi, (unsigned)m * 4u, 0, 0));
        u32x4_t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          unsigned w = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int qv = __float2int_rn(fminf(fmaxf(v[4 * k + b] * inv, -127.f), 127.f));
            w |= ((unsigned)qv & 0xffu) << (8 * b);
          }
          o[k] = w;
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, rsQ, (cok[0] && cok[1]) ? (unsigned)m * (unsigned)pp.ldq8 + (unsigned)ncol : OOB_OFFSET, 0, 0);
        if ((threadIdx.x & 48) == 0 && m < p.M) __hip_atomic_fetch_max(p.q8_amax + m, as_u32(am), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
    }
    if (mode == 1) {
      round_all();
#pragma unroll
      for (int c = 0; c < NV; ++c) v[c] *= gate[c];
    }
    {
      if (mode != 0 && use_res) {            // block-uniform
        round_all();
        const unsigned roff = (unsigned)m * (unsigned)pp.ldr * 2u + (unsigned)nbeg * 2u;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if constexpr (CH == 8) {
            const u32x4_t rv = __builtin_amdgcn_raw_buffer_load_b128(rsR, roff + c * 16, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[c * 8 + 2 * k] += bf_lo(rv[k]); v[c * 8 + 2 * k + 1] += bf_hi(rv[k]); }
          } else {
            const u32x2_t rv = __builtin_amdgcn_raw_buffer_load_b64(rsR, roff + c * 8, 0, 0);
            v[c * 4] += bf_lo(rv[0]); v[c * 4 + 1] += bf_hi(rv[0]); v[c * 4 + 2] += bf_lo(rv[1]); v[c * 4 + 3] += bf_hi(rv[1]);
          }
        }
      }
    }
    const unsigned ooff = (unsigned)m * (unsigned)ldo * 2u;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if constexpr (CH == 8) {
        u32x4_t o;
        o[0] = pack_bf2(v[c * 8], v[c * 8 + 1]); o[1] = pack_bf2(v[c * 8 + 2], v[c * 8 + 3]);
        o[2] = pack_bf2(v[c * 8 + 4], v[c * 8 + 5]); o[3] = pack_bf2(v[c * 8 + 6], v[c * 8 + 7]);
        __builtin_amdgcn_raw_buffer_store_b128(o, rsC, cok[c] ? ooff + coff[c] : OOB_OFFSET, 0, 0);   // (a sum with OOB_OFFSET would wrap)
      } else {
        u32x2_t o;
        o[0] = pack_bf2(v[c * 4], v[c * 4 + 1]); o[1] = pack_bf2(v[c * 4 + 2], v[c * 4 + 3]);
        __builtin_amdgcn_raw_buffer_store_b64(o, rsC, cok[c] ? ooff + coff[c] : OOB_OFFSET, 0, 0);
      }
    }
  };
  // two halves: the scheduler may hoist the residual loads of one half ahead of its arithmetic, not those of all WM rows
  // (which is what ran the 256x256 kernel out of registers)
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    row(i);
    if (WM >= 4 && i == WM / 2 - 1) __builtin_amdgcn_sched_barrier(0);
  }
}

// raw fp32 rows (attention scores of the VAE mid block): acc + bias, no bf16 rounding, no activation / gate / residual
template <int WM, int WN>
__device__ __forceinline__ void epilogue_f32(const TdGemmParams& pp, const ProbView& p, f32x4_t (&acc)[WN][WM], int mbeg, int nbeg) {
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(p.C, (unsigned)(((long long)(p.M - 1) * pp.ldc + pp.N) * 4));
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.bias, (unsigned)pp.N * 2u);
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const u32x2_t b = __builtin_amdgcn_raw_buffer_load_b64(rsB, (unsigned)(nbeg + j * 4) * 2u, 0, 0);
    const f32x4_t b4 = {bf_lo(b[0]), bf_hi(b[0]), bf_lo(b[1]), bf_hi(b[1])};
    const bool cok = nbeg + j * 4 + 4 <= pp.N;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const f32x4_t o4 = acc[j][i] + b4;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o4), rsC,
                                             cok ? (unsigned)(mbeg + i * 16) * (unsigned)pp.ldc * 4u + (unsigned)(nbeg + j * 4) * 4u : OOB_OFFSET, 0, 0);
    }
  }
}

}  // namespace

// FP8: both operands are OCP e4m3 bytes (K-contiguous, 128 elements = one 128-B LDS row per k-tile, so the staging
// and the swizzle are byte-for-byte those of the bf16 kernel); the two 16-B fragment reads of a lane feed ONE
// v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales; 2x the bf16 MFMA rate).  The k <-> (lane, byte) map of the
// instruction does not matter: both operands are read with the same map and the contraction sums over all k.
// Dequantisation is per output row (a_scale[m]) x per output column (w_scale[n]) on the fp32 accumulators.
// INT8 (I8): both operands are symmetric int8 bytes.  v_mfma_i32_16x16x64_i8 takes the SAME 16-byte fragments per lane and k-step as
// the bf16 instruction (64 int8 instead of 32 bf16 per k-step, twice the rate), so the int8 kernel IS the bf16 main loop -- staging,
// swizzle, W ring, instruction interleave -- with the other MFMA and k-tiles of 128 elements; the int32 accumulators are exact and
// become floats (x a_scale[m] x w_scale[n]) in front of the common epilogue.  Round 3: 8-bit operands whose quantisation noise is
// ~4x below e4m3's on Gaussian-like operands (uniform step max/127 against a 3-bit mantissa).
// One output tile: rows [m0, m0 + 32 WM) of problem `second_prob`, columns [n0, n0 + 64 WN).  Called once per workgroup.
template <int WM, int WN, bool CONV, bool FP8, bool I8>
__device__ __forceinline__ void gemm_tile(const TdGemmParams& p, char* smem, const bool second_prob, const int m0, const int n0) {
  constexpr int BM = 32 * WM, BN = 64 * WN;
  constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES;
  constexpr int GA = BM / 8, GW = BN / 8;           // 8-row staging groups per tile
  constexpr int SA = (GA + 7) / 8, SW = (GW + 7) / 8;  // staging instructions per wave
  constexpr int NV = 4 * WN;                          // contiguous output columns per lane
  constexpr unsigned ESZ = (FP8 || I8) ? 1u : 2u;     // operand element size in bytes
  static_assert(!(FP8 || I8) || !CONV, "no 8-bit convolution");
  static_assert(!(FP8 && I8), "one operand type");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  const bf16_t* Aptr = second_prob ? p.g_A : p.A;
  const bf16_t* Wptr = second_prob ? p.g_W : p.W;
  const int part = p.k_parts > 1 ? (int)blockIdx.y : 0;      // split-K launches (bf16, one problem): this workgroup contracts K-range `part`
  if (p.k_parts > 1) { Aptr += (size_t)part * p.K; Wptr += (size_t)part * p.K; }
  ProbView pv;
  pv.bias = second_prob ? p.g_bias : p.bias;
  pv.gate = second_prob ? p.g_gate : p.gate;
  pv.res = second_prob ? p.g_res : p.res;
  pv.C = second_prob ? p.g_C : p.C;
  pv.M = second_prob ? p.g_M : p.M;
  if (p.k_parts > 1) pv.C = (bf16_t*)((float*)p.C + (size_t)part * p.M * p.ldc);      // (out_f32: its slab of the fp32 partial sums)
  pv.q8 = second_prob ? p.g_q8 : p.q8; pv.q8_inv = second_prob ? p.g_q8_inv : p.q8_inv; pv.q8_amax = second_prob ? p.g_q8_amax : p.q8_amax;
  pv.q8_smooth = second_prob ? p.g_q8_smooth : p.q8_smooth;
  if (m0 >= pv.M) return;      // (a sub-tile of a split tail tile that lies wholly below the problem's last row; workgroup-uniform)

  // ---- buffer descriptors (wave-uniform; OOB rows read as zero) -----------------------------
  const unsigned bytesA = CONV ? (unsigned)((long long)(p.conv_H >> p.conv_up) * (p.conv_W >> p.conv_up) * p.conv_Cin * 2)
                                : (unsigned)(((long long)(pv.M - 1) * p.lda + p.K) * ESZ);
  const unsigned bytesW = (unsigned)((((long long)p.N - 1) * p.ldw + p.K) * ESZ);
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Aptr, 0, bytesA, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wptr, 0, bytesW, 0x00020000);

  // ---- per-lane staging source offsets ------------------------------------------------------
  const int srow = lane >> 3;                         // row inside the 8-row group
  const int schunk = ((lane & 7) ^ srow) << 4;        // swizzled 16-B source chunk
  // NS = SA + SW staging instructions per wave per k-tile; instruction s < SA moves an 8-row group of the
  // A tile, the rest of the W tile.  A wave whose group index falls past the tile (tile rows not a
  // multiple of 64) re-stages another wave's group instead: identical bytes, and no divergent skip, so the
  // instructions can be spread through the MFMA stream of the main loop.
  constexpr int NS = SA + SW;
  unsigned voffS[NS];
  int ldsS[NS];
#pragma unroll
  for (int s = 0; s < SA; ++s) {
    int g = wid + 8 * s;
    if (GA % 8 != 0 && g >= GA) g %= GA;
    const int row = m0 + g * 8 + srow;
    voffS[s] = (unsigned)row * (unsigned)p.lda * ESZ + schunk;
    ldsS[s] = g * 1024;
  }
#pragma unroll
  for (int s = 0; s < SW; ++s) {
    int g = wid + 8 * s;
    if (GW % 8 != 0 && g >= GW) g %= GW;
    const int rho = g * 8 + srow;                     // LDS row of the block's W slab
    const int wcol = rho / (16 * WN);
    const int rem = rho - wcol * (16 * WN);
    const int j = rem >> 4, i16 = rem & 15;
    const int n = n0 + wcol * (16 * WN) + (i16 >> 2) * NV + j * 4 + (i16 & 3);
    voffS[SA + s] = (unsigned)n * (unsigned)p.ldw * ESZ + schunk;
    ldsS[SA + s] = g * 1024;
  }
  // Implicit-GEMM 3x3 convolution (CONV): A is an NHWC image [Hin*Win, Cin]; k-tile kt covers tap kt / (Cin/64)
  // and input channels 64*(kt % (Cin/64)).  A row (= output pixel) reads input pixel (y+dy, x+dx) -- or its
  // nearest-neighbour parent when the conv follows a 2x upsample -- and taps outside the image are sent past the
  // descriptor range, so the zero padding costs nothing.
  int cy[SA], cx[SA];
  if constexpr (CONV) {
#pragma unroll
    for (int s = 0; s < SA; ++s) {
      int g2 = wid + 8 * s;
      if (GA % 8 != 0 && g2 >= GA) g2 %= GA;
      const int pix = m0 + g2 * 8 + srow;
      cy[s] = pix < pv.M ? pix / p.conv_W : -4;   // rows past M never validate
      cx[s] = pix - (pix / p.conv_W) * p.conv_W;
    }
  }
  auto conv_voff = [&](int s, int kt) -> unsigned {
    const int cpt = p.conv_Cin >> 6;
    const int tap = kt / cpt, c0 = (kt - tap * cpt) << 6;
    const int dy = (tap * 11 >> 5) - 1, dx = tap - (tap * 11 >> 5) * 3 - 1;
    const int yy = cy[s] + dy, xx = cx[s] + dx;
    const bool ok = (unsigned)yy < (unsigned)p.conv_H && (unsigned)xx < (unsigned)p.conv_W;
    const int sy = yy >> p.conv_up, sx = xx >> p.conv_up, win = p.conv_W >> p.conv_up;
    const unsigned off = ((unsigned)(sy * win + sx) * (unsigned)p.conv_Cin + (unsigned)c0) * 2u + schunk;
    return ok ? off : 0xFFFFFF00u;
  };
  // LDS: [A buf 0 | A buf 1 | W buf 0 | W buf 1 | W buf 2]
  constexpr int W_REGION = 2 * A_BYTES;
  auto stage_one = [&](int s, int abuf, int wbuf, int kt) {
    if (s < SA) {
      if constexpr (CONV)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (TD_LDS void*)(smem + abuf * A_BYTES + ldsS[s]), 16, conv_voff(s, kt), 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (TD_LDS void*)(smem + abuf * A_BYTES + ldsS[s]), 16, voffS[s], kt * (BK * 2), 0, 0);
    } else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (TD_LDS void*)(smem + W_REGION + wbuf * W_BYTES + ldsS[s]), 16, voffS[s], kt * (BK * 2), 0, 0);
  };

  // ---- fragment read offsets ------------------------------------------------------------------
  const int frow = lane & 15;
  const int foff0 = frow * ROW_BYTES + ((((lane >> 4)) ^ (lane & 7)) << 4);  // k-step 0; k-step 1 = ^64
  const int aoff = (wr * 16 * WM) * ROW_BYTES;
  const int woff = (wc * 16 * WN) * ROW_BYTES;

  f32x4_t acc[WN][WM];
#pragma unroll
  for (int j = 0; j < WN; ++j)
#pragma unroll
    for (int i = 0; i < WM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nt = p.probe == 2 ? 1 : p.K / ((FP8 || I8) ? 2 * BK : BK);      // a k-tile is one 128-B row: 64 bf16 or 128 fp8 / int8
#pragma unroll
  for (int s = 0; s < NS; ++s) stage_one(s, 0, 0, 0);
  if constexpr (FP8) {
    typedef __attribute__((ext_vector_type(4))) int i32x4_t;
    typedef __attribute__((ext_vector_type(8))) int i32x8_t;
    auto frag = [&](const char* base) -> i32x8_t {   // the lane's 16-B chunks of both 64-B halves of the row
      const i32x4_t lo = *(const i32x4_t*)(base + foff0), hi = *(const i32x4_t*)(base + (foff0 ^ 64));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
#pragma unroll
    for (int s = SA; s < NS; ++s) stage_one(s, 0, 1, min(1, nt - 1));
    int wcur = 0;
    constexpr int MASK_VMEM = 0x010, MASK_DS_READ = 0x100, MASK_MFMA = 0x008;
    constexpr int S_PER_IT = (NS + WM - 1) / WM;
    for (int t = 0; t < nt; ++t) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SW) : "memory");
      __builtin_amdgcn_s_barrier();
      const char* wb = smem + W_REGION + wcur * W_BYTES + woff;
      const char* ab = smem + (t & 1) * A_BYTES + aoff;
      const int kt_a = min(t + 1, nt - 1), kt_w = min(t + 2, nt - 1);
      const int abuf_next = (t + 1) & 1;
      const int wbuf_next = wcur == 0 ? 2 : wcur - 1;
      wcur = wcur == 2 ? 0 : wcur + 1;
      i32x8_t wf[WN], af[2];
      wf[0] = frag(wb);
      af[0] = frag(ab);
      __builtin_amdgcn_sched_group_barrier(MASK_DS_READ, 4, 0);
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int cur = i & 1;
        int nst = 0;
#pragma unroll
        for (int q = 0; q < S_PER_IT; ++q) {
          const int s = i * S_PER_IT + q;
          if (s < NS) { stage_one(s, abuf_next, wbuf_next, s < SA ? kt_a : kt_w); ++nst; }
        }
        if (i == 0) {
#pragma unroll
          for (int j = 1; j < WN; ++j) wf[j] = frag(wb + j * 16 * ROW_BYTES);
        }
        if (i + 1 < WM) af[cur ^ 1] = frag(ab + (i + 1) * 16 * ROW_BYTES);
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[cur], acc[j][i], 0, 0, 0, 127, 0, 127);
        switch (nst) {
          case 1: __builtin_amdgcn_sched_group_barrier(MASK_VMEM, 1, 0); break;
          case 2: __builtin_amdgcn_sched_group_barrier(MASK_VMEM, 2, 0); break;
          case 3: __builtin_amdgcn_sched_group_barrier(MASK_VMEM, 3, 0); break;
          case 4: __builtin_amdgcn_sched_group_barrier(MASK_VMEM, 4, 0); break;
          case 5: __builtin_amdgcn_sched_group_barrier(MASK_VMEM, 5, 0); break;
          default: break;
        }
        if (i == 0) {
          // the remaining W fragments arrive one MFMA ahead of their use; the next A fragment behind the last
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            if (j + 1 < WN) __builtin_amdgcn_sched_group_barrier(MASK_DS_READ, 2, 0);
            else if (WM > 1) __builtin_amdgcn_sched_group_barrier(MASK_DS_READ, 2, 0);
            __builtin_amdgcn_sched_group_barrier(MASK_MFMA, 1, 0);
          }
        } else {
          if (i + 1 < WM) __builtin_amdgcn_sched_group_barrier(MASK_DS_READ, 2, 0);
          __builtin_amdgcn_sched_group_barrier(MASK_MFMA, WN, 0);
        }
      }
    }
  } else
  if (WM >= 8 && !CONV && pv.M - m0 <= p.ragged_rows) {      // (the big tiles only: 64 rows are at most four m-tiles of the lower wave row, none of the upper)
  // Ragged tile: at most 64 of the tile's rows exist (config 5's 258 text rows and 4 354 joint rows are whole 256-row tiles plus TWO rows:
  // 5.6 % of that shape's tiles).  The staging protocol is the main loop's (same DMA instructions in the same order, same counted wait --
  // every wave takes part), but only the m-tiles that hold rows are multiplied: the waves of the upper half have none, the others at most
  // four, so the tile costs its operand traffic instead of a full tile of MFMAs on zeros.  Plain code: nothing here is matrix-bound.
  const int mt_valid = wr == 0 ? (pv.M - m0 + 15) >> 4 : 0;      // wave-uniform
#pragma unroll
  for (int s = SA; s < NS; ++s) stage_one(s, 0, 1, min(1, nt - 1));
  int wcur = 0;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SW) : "memory");
    __builtin_amdgcn_s_barrier();
    const char* wb = smem + W_REGION + wcur * W_BYTES + woff;
    const char* ab = smem + (t & 1) * A_BYTES + aoff;
    const int kt_a = min(t + 1, nt - 1), kt_w = min(t + 2, nt - 1);
    const int abuf_next = (t + 1) & 1;
    const int wbuf_next = wcur == 0 ? 2 : wcur - 1;
    wcur = wcur == 2 ? 0 : wcur + 1;
#pragma unroll
    for (int s = 0; s < NS; ++s) stage_one(s, abuf_next, wbuf_next, s < SA ? kt_a : kt_w);
    if (mt_valid > 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int fo = foff0 ^ (ks << 6);
        bf16x8_t wf[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) wf[j] = *(const bf16x8_t*)(wb + j * 16 * ROW_BYTES + fo);
#pragma unroll
        for (int i = 0; i < (WM < 4 ? WM : 4); ++i) {
          if (i < mt_valid) {
            const bf16x8_t af = *(const bf16x8_t*)(ab + i * 16 * ROW_BYTES + fo);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              if constexpr (I8) {
                typedef __attribute__((ext_vector_type(4))) int i32x4_t;
                acc[j][i] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_t, wf[j]), __builtin_bit_cast(i32x4_t, af),
                                                                                               __builtin_bit_cast(i32x4_t, acc[j][i]), 0, 0, 0));
              } else {
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af, acc[j][i], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  }
  } else
  {
  // Main loop, one barrier per k-tile.  Inside a tile every instruction kind is spread through the MFMA
  // stream (pinned with sched_group_barrier, hipcc otherwise clusters them):
  //  * the NS LDS-DMA instructions of tile t+1 ride on the first m-tiles (their ~60-cycle issue cost hides
  //    behind MFMAs instead of fronting the tile while the matrix pipe idles);
  //  * the WN W-fragments of a k-step stay resident, A-fragments are read one m-tile ahead of the MFMAs
  //    that consume them, the next k-step's W-fragments behind the last m-tiles.
  //  * the W tile (the operand that streams cold from HBM: every layer has its own weights)
  //    is prefetched TWO k-tiles ahead into a 3-deep W ring, the A tile (L2-resident activations) one
  //    ahead.  A DMAs are issued before the W DMAs of an iteration, so the counted `s_waitcnt vmcnt(SW)`
  //    at the barrier retires tile t+1's operands and leaves the W(t+2) transfers in flight across it.
#pragma unroll
  for (int s = SA; s < NS; ++s) stage_one(s, 0, 1, min(1, nt - 1));
  int wcur = 0;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SW) : "memory");
    __builtin_amdgcn_s_barrier();
    const int wslot = wcur;
    const char* wb = smem + W_REGION + wslot * W_BYTES + woff;
    const char* ab = smem + (t & 1) * A_BYTES + aoff;
    // tiles past the last are harmless re-loads of the last one (keeps the loop body branch-free)
    const int kt_a = min(t + 1, nt - 1);
    const int kt_w = min(t + 2, nt - 1);
    const int abuf_next = (t + 1) & 1;
    const int wbuf_next = wcur == 0 ? 2 : wcur - 1;   // (wcur + 2) % 3
    wcur = wcur == 2 ? 0 : wcur + 1;
    bf16x8_t wf[2][WN], af[2];
#pragma unroll
    for (int j = 0; j < WN; ++j) wf[0][j] = *(const bf16x8_t*)(wb + j * 16 * ROW_BYTES + foff0);
    af[0] = *(const bf16x8_t*)(ab + foff0);
    __builtin_amdgcn_sched_group_barrier(0x100, WN + 1, 0);  // the k-step-0 fragments first
    constexpr int MASK_VMEM = 0x010, MASK_DS_READ = 0x100, MASK_MFMA = 0x008;
    constexpr int NIT = 2 * WM;                       // (k-step, m-tile) iterations per tile
    constexpr int S_PER_IT = (NS + NIT - 1) / NIT;    // staging instructions per iteration
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int fo = foff0 ^ (ks << 6);
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int it = ks * WM + i;
        const int cur = it & 1;
        int nst = 0;
#pragma unroll
        for (int q = 0; q < S_PER_IT; ++q) {
          const int s = it * S_PER_IT + q;
          if (s < NS) { stage_one(s, abuf_next, wbuf_next, s < SA ? kt_a : kt_w); ++nst; }
        }
        // prefetch the next A fragment (next m-tile, or m-tile 0 of the next k-step)
        if (i + 1 < WM) af[cur ^ 1] = *(const bf16x8_t*)(ab + (i + 1) * 16 * ROW_BYTES + fo);
        else if (ks == 0) af[cur ^ 1] = *(const bf16x8_t*)(ab + (fo ^ 64));
        // prefetch the next k-step's W fragments behind the last m-tiles (fragment j rides on m-tile WM-1-j%WM)
        int nwf = 0;
        if (ks == 0) {
#pragma unroll
          for (int j = 0; j < WN; ++j)
            if (WM - 1 - (j % WM) == i) {
              wf[1][j] = *(const bf16x8_t*)(wb + j * 16 * ROW_BYTES + (fo ^ 64));
              ++nwf;
            }
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          if constexpr (I8) {
            typedef __attribute__((ext_vector_type(4))) int i32x4_t;
            acc[j][i] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_t, wf[ks][j]), __builtin_bit_cast(i32x4_t, af[cur]),
                                                                                           __builtin_bit_cast(i32x4_t, acc[j][i]), 0, 0, 0));
          } else {
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], af[cur], acc[j][i], 0, 0, 0);
          }
        }
        const int nreads = ((i + 1 < WM || ks == 0) ? 1 : 0) + nwf;
        switch (nst) {  // the builtin wants literal counts
          case 1: __builtin_amdgcn_sched_group_barrier(MASK_VMEM, 1, 0); break;
          case 2: __builtin_amdgcn_sched_group_barrier(MASK_VMEM, 2, 0); break;
          case 3: __builtin_amdgcn_sched_group_barrier(MASK_VMEM, 3, 0); break;
          default: break;
        }
        switch (nreads) {
          case 1: __builtin_amdgcn_sched_group_barrier(MASK_DS_READ, 1, 0); break;
          case 2: __builtin_amdgcn_sched_group_barrier(MASK_DS_READ, 2, 0); break;
          case 3: __builtin_amdgcn_sched_group_barrier(MASK_DS_READ, 3, 0); break;
          case 4: __builtin_amdgcn_sched_group_barrier(MASK_DS_READ, 4, 0); break;
          case 5: __builtin_amdgcn_sched_group_barrier(MASK_DS_READ, 5, 0); break;
          default: break;
        }
        __builtin_amdgcn_sched_group_barrier(MASK_MFMA, WN, 0);
      }
    }
  }

  }

  // ---- epilogue: lane owns NV contiguous columns of row (lane&15) of each m-tile -------------
  const int nbeg = n0 + wc * 16 * WN + (lane >> 4) * NV;
  if (nbeg >= p.N || p.probe == 1) return;      // (probe 1: a run-time condition, so the k-loop stays; nothing else is added to the kernel)
  const bool second = (p.C2 != nullptr) && (n0 >= p.n_split);
  const int act = second ? p.act2 : p.act;
  const int mbeg = m0 + wr * 16 * WM + frow;
  if constexpr (FP8 || I8) {   // y = (sum q_a q_w) * a_scale[row] * w_scale[col]
    // range-checked loads: rows >= M / columns >= N read a zero scale (their outputs are dropped anyway)
    const __amdgpu_buffer_rsrc_t rsSa = make_rsrc(second_prob ? p.g_a_scale : p.a_scale, (unsigned)pv.M * 4u);
    const __amdgpu_buffer_rsrc_t rsSw = make_rsrc(second_prob ? p.g_w_scale : p.w_scale, (unsigned)p.N * 4u);
    float swv[NV];
#pragma unroll
    for (int c = 0; c < NV; c += 4) {
      const f32x4_t w4 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsSw, (unsigned)(nbeg + c) * 4u, 0, 0));
      swv[c] = w4[0]; swv[c + 1] = w4[1]; swv[c + 2] = w4[2]; swv[c + 3] = w4[3];
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const float sr = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsSa, (unsigned)(mbeg + i * 16) * 4u, 0, 0));
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (I8) acc[j][i][r] = (float)(int)as_u32(acc[j][i][r]) * (sr * swv[j * 4 + r]);     // exact int32 sum -> float
          else acc[j][i][r] *= sr * swv[j * 4 + r];
        }
    }
  }
  // (an activation followed by a gate / residual does not occur on this path: act wins)
  if (p.out_f32) {
    epilogue_f32<WM, WN>(p, pv, acc, mbeg, nbeg);
    return;
  }
  const int mode = act != TD_ACT_NONE ? 0 : (pv.gate ? 1 : (pv.res ? 2 : 0));
  epilogue<WM, WN, I8 && WN == 4>(p, pv, acc, mbeg, nbeg, second, act, mode);
}

// XCD-aware numbering: workgroup ids go round-robin over the 8 XCDs; logical index t gives every XCD one CONTIGUOUS run of tiles (its 32 resident
// tiles then share A / W panels through that XCD's L2), in dispatch order inside the run.
__device__ __forceinline__ int xcd_contiguous(const int bid, const int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}


// TAIL > 1 (256-row tiles, no conv): the launcher found that the tile count leaves a last, mostly empty round of the chip's CUs
// (816 tiles on 256 CUs = 3.19 rounds: the launch takes 4 tile times), and cut the LAST `tail_tiles` tiles -- the ones dispatched
// behind the full rounds -- into TAIL sub-tiles of 32 WM / TAIL rows each: workgroups [tail_first_wg, grid) take one sub-tile, so the
// last round costs 1 / TAIL of a tile time (x the smaller tile's lower efficiency) instead of a whole one.  No split along K, no
// fix-up pass: every output element is still produced by one workgroup with the full contraction, bit-identical to the unsplit launch.
template <int WM, int WN, bool CONV = false, bool FP8 = false, bool I8 = false, int TAIL = 1>
__global__ __launch_bounds__(512, 1) void td_gemm_bf16_nt_kernel(const TdGemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // body uses gfx950-only types (__amdgpu_buffer_rsrc_t): the host pass only needs the stub
  static_assert(TAIL == 1 || (WM % TAIL == 0 && !CONV), "tail sub-tiles cut the m extent of the tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 32 * WM, BN = 64 * WN;
  int t, sub = 0;
  bool tail = false;
  if (TAIL > 1 && (int)blockIdx.x >= p.tail_first_wg) {
    const int u = xcd_contiguous((int)blockIdx.x - p.tail_first_wg, (int)gridDim.x - p.tail_first_wg);
    t = p.tail_first_wg + u / TAIL;      // (full tiles: one workgroup each, so the first tail tile's logical index is tail_first_wg)
    sub = u - (u / TAIL) * TAIL;
    tail = true;
  } else {
    t = xcd_contiguous((int)blockIdx.x, TAIL > 1 ? p.tail_first_wg : (int)gridDim.x);
  }
  // grouped M ordering of the logical index
  constexpr int GROUP_M = 4;
  const int per_group = GROUP_M * p.tiles_n;
  const int gid = t / per_group;
  const int first_m = gid * GROUP_M;
  const int gsize = min(p.tiles_m - first_m, GROUP_M);
  const int in_g = t - gid * per_group;
  int tm = first_m + in_g % gsize;
  const int tn = in_g / gsize;
  // grouped launch: m-tiles [0, tiles_m0) belong to problem 0, the rest to problem 1 (same N, K, strides)
  const bool second_prob = tm >= p.tiles_m0;
  if (second_prob) tm -= p.tiles_m0;
  if constexpr (TAIL > 1) {
    if (tail) {
      gemm_tile<WM / TAIL, WN, CONV, FP8, I8>(p, smem, second_prob, tm * BM + sub * (BM / TAIL), tn * BN);
      return;
    }
  }
  gemm_tile<WM, WN, CONV, FP8, I8>(p, smem, second_prob, tm * BM, tn * BN);
#endif
}

namespace {

// How many CUs the device has (the round size of a one-workgroup-per-CU grid); cached per device.
int cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int v = cached[dev & 63].load(std::memory_order_acquire);
  if (v <= 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cached[dev & 63].store(v, std::memory_order_release);
  }
  return v;
}

// Tail split of a grid of `tiles` equal tiles on `cus` CUs (one workgroup per CU): the launch takes ceil(tiles / cus) tile times.  Cutting the
// last `rem = tiles mod cus` tiles into s sub-tiles makes it floor(tiles / cus) + ceil(rem s / cus) / s x (what the smaller tile loses in
// efficiency: measured ~8 % at s = 2, ~15 % at s = 4).  Returns the best s in {1, 2, 4} -- 1 unless a split saves at least 4 % of the launch.
//
// OFF BY DEFAULT (TD_GEMM_TAIL=auto|2|4 turns it on).  A lone cold-weight launch gains what the model says (q|k|v -12 %, ff.net.0 -14 % bf16; -7 / -8 %
// int8, profiles/r4_gemm_probe.log) -- and the whole image LOSES: same box, same process, alternated three times, one image in flight: bf16 0.5783 with
// the split vs 0.5846 images/s without, the 8-bit policy 0.907 vs 0.922; two in flight 0.580 vs 0.583 / 0.964 vs 0.968
// (profiles/r4_tail_split_ab.log).  On real data this chip runs the denoise loop at a clock set by its power budget: an idle end of a round is
// power the other rounds get back as clock, while the sub-tiles that fill it move 1.5-2.5 x the operand bytes per FLOP (the guide's rule 28:
// what pays is energy per FLOP, not occupancy).  Kept as a launch form with its bit-identity tests because the conclusion depends on the regime
// (a device that is not power-limited on this loop would gain the cold-launch figure).
int tail_split(long long tiles, int cus) {
  const char* f = getenv("TD_GEMM_TAIL");      // (read per launch so one process can time every form)
  if (!f || getenv("TD_GEMM_NO_TAIL")) return 1;
  if (atoi(f) == 2 || atoi(f) == 4) return atoi(f);
  if (strcmp(f, "auto") != 0) return 1;
  const long long full = tiles / cus, rem = tiles % cus;
  if (rem == 0 || full == 0) return 1;          // (less than one round: the tile chooser already picks a smaller tile there)
  const double base = (double)(full + 1);
  double best = base * 0.96;
  int pick = 1;
  for (int s2 = 2; s2 <= 4; s2 *= 2) {
    const double cost = (double)full + (double)((rem * s2 + cus - 1) / cus) / s2 * (s2 == 2 ? 1.08 : 1.15);
    if (cost < best) { best = cost; pick = s2; }
  }
  return pick;
}

template <int WM, int WN, bool CONV = false, bool FP8 = false, bool I8 = false, int TAIL = 1>
int launch_cfg(const TdGemmParams& p0, hipStream_t stream) {
  constexpr int BM = 32 * WM, BN = 64 * WN;
  constexpr int LDS = (2 * BM + 3 * BN) * ROW_BYTES;
  TdGemmParams p = p0;
  p.tiles_m0 = (p.M + BM - 1) / BM;
  p.tiles_m = p.tiles_m0 + (p.g_M > 0 ? (p.g_M + BM - 1) / BM : 0);
  p.tiles_n = (p.N + BN - 1) / BN;
  p.ragged_rows = getenv("TD_GEMM_NO_RAGGED") ? 0 : 64;      // (A/B switch of the ragged-tile loop; read per launch so one process can time both)
  if (const char* pr = getenv("TD_GEMM_PROBE")) p.probe = atoi(pr);
  if (p.C2) TD_CHECK_ARG(p.n_split % BN == 0, "td_gemm: n_split=%d must be a multiple of the N tile %d", p.n_split, BN);
  int grid = p.tiles_m * p.tiles_n;
  if constexpr (TAIL == 1 && WM == 8 && WN == 4 && !CONV) {      // the 256 x 256 tile of the block Linears: see whether its last round is worth cutting up
    const int s = (p.no_tail || p.k_parts > 1) ? 1 : tail_split(grid, cu_count());
    if (s == 2) return launch_cfg<WM, WN, CONV, FP8, I8, 2>(p0, stream);
    if (s == 4) return launch_cfg<WM, WN, CONV, FP8, I8, 4>(p0, stream);
  }
  if constexpr (TAIL > 1) {
    const int cus = cu_count();
    p.tail_first_wg = grid / cus * cus;                        // a multiple of 8: the XCD numbering of both parts stays aligned
    grid = p.tail_first_wg + (grid - p.tail_first_wg) * TAIL;
  }
  // the dynamic-LDS limit is a per-device function attribute: set it once per device (a process may drive several)
  static std::atomic<unsigned long long> attr_done{0ull};
  int dev = 0;
  TD_CHECK_HIP(hipGetDevice(&dev));
  if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
    TD_CHECK_HIP(hipFuncSetAttribute((const void*)td_gemm_bf16_nt_kernel<WM, WN, CONV, FP8, I8, TAIL>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
  }
  if (p.ldw == 0) p.ldw = p.K;
  hipLaunchKernelGGL((td_gemm_bf16_nt_kernel<WM, WN, CONV, FP8, I8, TAIL>), dim3(grid, p.k_parts > 1 ? p.k_parts : 1), dim3(512), LDS, stream, p);
  TD_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// ---- K split over workgroups (TdGemmParams::split_k) --------------------------------------------------------------------------
// Second launch of a split GEMM: out[m, n] = epilogue(sum over parts, in index order, of the fp32 partial sums) with the tile kernel's rounding
// points (Linear output + bias; a residual is added to the bf16-rounded output; the final pack rounds).  4 columns per thread.
__global__ __launch_bounds__(256) void td_gemm_splitk_reduce_kernel(const float* __restrict__ ws, const int parts, const int M, const int N, const bf16_t* bias,
                                                                    const bf16_t* res, const int ldr, bf16_t* C, const int ldc, bf16_t* C2, const int ldc2, const int n_split) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = N >> 2;
  if (idx >= (long long)M * n4) return;
  const int m = (int)(idx / n4), n = (int)(idx - (long long)m * n4) * 4;
  const size_t slab = (size_t)M * N;
  const float* src = ws + (size_t)m * N + n;
  f32x4_t v = *(const f32x4_t*)src;
  for (int k = 1; k < parts; ++k) v += *(const f32x4_t*)(src + (size_t)k * slab);
  if (bias) {
    const u32x2_t b = *(const u32x2_t*)(bias + n);
    v += f32x4_t{bf_lo(b[0]), bf_hi(b[0]), bf_lo(b[1]), bf_hi(b[1])};
  }
  if (res) {
    const u32x2_t r = *(const u32x2_t*)(res + (size_t)m * ldr + n);
    v = f32x4_t{rbf(v[0]) + bf_lo(r[0]), rbf(v[1]) + bf_hi(r[0]), rbf(v[2]) + bf_lo(r[1]), rbf(v[3]) + bf_hi(r[1])};
  }
  const u32x2_t o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
  if (C2 && n >= n_split) *(u32x2_t*)(C2 + (size_t)m * ldc2 + (n - n_split)) = o;
  else *(u32x2_t*)(C + (size_t)m * ldc + n) = o;
}

// The same reduction followed by the RMSNorm of the row it has just finished (TdGemmParams::sk_norm_w): one workgroup of NCH waves per row, wave c
// owning the 512-column chunk c with td_norm_rows_kernel's lane map (lane l: columns 512 c + 8 l .. + 7).  That kernel adds a lane's squares chunk
// after chunk in ONE chain before the cross-lane sum; here the chunks' bf16-rounded values meet in LDS and every wave walks the same chain, so the
// normalised row is bit-identical to the one the separate norm launch would have produced from the bf16 row written here -- while the partial sums
// are fetched by NCH waves at once instead of one (one wave per row: 9.8 us at N = 1536, 19 us at N = 3584 for 256 rows, more than the two launches
// it replaced).
template <int NCH>
__global__ __launch_bounds__(NCH * 64) void td_gemm_splitk_reduce_norm_kernel(const float* __restrict__ ws, const int parts, const int M, const bf16_t* bias, const bf16_t* res,
                                                                              const int ldr, bf16_t* C, const int ldc, const bf16_t* nw, bf16_t* nout, const int nld, const float eps) {
  constexpr int N = NCH * 512;
  __shared__ float xs[NCH][8][64];      // [chunk][element][lane]: conflict-free for the lane-contiguous accesses below
  const int lane = threadIdx.x & 63, c = threadIdx.x >> 6;
  const int m = blockIdx.x;
  const size_t slab = (size_t)M * N;
  const int n = c * 512 + lane * 8;
  float x[8];
  {
    const float* src = ws + (size_t)m * N + n;
    f32x4_t a = *(const f32x4_t*)src, b = *(const f32x4_t*)(src + 4);
    for (int k = 1; k < parts; ++k) { a += *(const f32x4_t*)(src + (size_t)k * slab); b += *(const f32x4_t*)(src + (size_t)k * slab + 4); }
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    if (bias) {
      const u32x4_t bb = *(const u32x4_t*)(bias + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[2 * i] += bf_lo(bb[i]); v[2 * i + 1] += bf_hi(bb[i]); }
    }
    if (res) {
      const u32x4_t r = *(const u32x4_t*)(res + (size_t)m * ldr + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[2 * i] = rbf(v[2 * i]) + bf_lo(r[i]); v[2 * i + 1] = rbf(v[2 * i + 1]) + bf_hi(r[i]); }
    }
    const u32x4_t o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    *(u32x4_t*)(C + (size_t)m * ldc + n) = o;
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[2 * i] = bf_lo(o[i]); x[2 * i + 1] = bf_hi(o[i]); }
#pragma unroll
    for (int i = 0; i < 8; ++i) xs[c][i][lane] = x[i];
  }
  __syncthreads();
  // td_norm_rows_kernel, rms form: a lane's squares are added chunk by chunk, element by element, then across the wave
  float sq = 0.f;
#pragma unroll
  for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float t = xs[cc][i][lane]; sq += t * t; }
  const float rstd = rsqrtf(wave_sum(sq) * (1.0f / N) + eps);
  const u32x4_t wr = *(const u32x4_t*)(nw + n);
  float y[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    y[2 * i] = rbf(rbf(x[2 * i] * rstd) * bf_lo(wr[i]));
    y[2 * i + 1] = rbf(rbf(x[2 * i + 1] * rstd) * bf_hi(wr[i]));
  }
  *(u32x4_t*)(nout + (size_t)m * nld + n) = u32x4_t{pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3]), pack_bf2(y[4], y[5]), pack_bf2(y[6], y[7])};
}

namespace {

constexpr long long SPLITK_POOL_BYTES = 64ll << 20;

// one pooled buffer of partial sums per (device, stream): launches on one stream never overlap
int gemm_splitk_pool(hipStream_t stream, float** out) {
  struct Entry { int dev; hipStream_t stream; float* ws; };
  static std::mutex mu;
  static std::vector<Entry> pool;
  int dev = 0;
  TD_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  for (auto& e : pool)
    if (e.dev == dev && e.stream == stream) { *out = e.ws; return 0; }
  float* w = nullptr;
  TD_CHECK_HIP(hipMalloc((void**)&w, (size_t)SPLITK_POOL_BYTES));
  pool.push_back(Entry{dev, stream, w});
  *out = w;
  return 0;
}

// Parts of a split launch, or 1.  Under -1: as many as spread tiles x parts over the CUs, at most 16, at least 4 k-tiles each, a divisor of the
// k-tile count, partial sums inside the workspace.
int plan_split_k(const TdGemmParams& p, int cfg) {
  if (p.split_k == 0 || p.split_k == 1) return 1;
  const bool covered = !p.fp8 && !p.i8 && p.conv_H == 0 && p.g_M == 0 && !p.out_f32 && p.act == TD_ACT_NONE && p.act2 == TD_ACT_NONE && !p.gate && !p.q8 && !p.glu_I;
  if (!covered) return 1;
  static const int bm[5] = {256, 256, 32, 288, 256}, bn[5] = {256, 64, 256, 192, 128};
  const long long tiles = (long long)((p.M + bm[cfg] - 1) / bm[cfg]) * ((p.N + bn[cfg] - 1) / bn[cfg]);
  const int ktiles = p.K / 64;
  const long long cap = p.sk_ws ? p.sk_ws_bytes : SPLITK_POOL_BYTES;
  const long long by_ws = cap / ((long long)p.M * p.N * 4);
  if (p.split_k > 1) return (ktiles % p.split_k == 0 && p.split_k <= by_ws) ? p.split_k : 1;      // (1 is then refused by the caller)
  int limit = (int)std::min<long long>(std::min<long long>(16, cu_count() / std::max<long long>(tiles, 1)), ktiles / 4);
  limit = (int)std::min<long long>(limit, by_ws);
  for (int d = limit; d >= 2; --d)
    if (ktiles % d == 0) return d;
  return 1;
}

// 64 < M <= 256 with nothing forced (the decode step of 65-256 sequences: every Linear is one row of tiles): tile width and parts are chosen
// TOGETHER from a cost model of the launch -- rounds of the CUs x (k-tiles per part x the tile's k-tile time + a fixed part) + the reduction's
// traffic -- over the 64-, 128- and 256-column tiles.  What it fixes: the 2B decoder's gate | up (N = 17 920) is 280 tiles of 64 columns, 1.09
// rounds of 256 CUs, i.e. two (55 us); as 140 tiles of 128 columns it is one.  Microseconds per k-tile measured on the decode shapes (rocprofv3).
struct WidePlan { int cfg, parts; };
WidePlan plan_wide(const TdGemmParams& p) {
  static const int cfgs[3] = {1, 4, 0}, bns[3] = {64, 128, 256};
  static const double ck[3] = {0.45, 0.6, 1.1};
  const int cus = cu_count(), ktiles = p.K / 64;
  WidePlan best{td_gemm_config_id(p.M, p.N, p.K), 1};
  double best_cost = 1e30;
  for (int i = 0; i < 3; ++i) {
    const int parts = plan_split_k(p, cfgs[i]);
    if (parts == 1 && p.C2 && p.n_split % bns[i] != 0) continue;
    const long long wgs = (long long)((p.N + bns[i] - 1) / bns[i]) * parts;
    const double rounds = (double)((wgs + cus - 1) / cus);
    const double cost = rounds * ((double)(ktiles / parts) * ck[i] + 3.0) + (parts > 1 ? 4.0 + (parts + 0.5) * (double)p.M * p.N * 4.0 / 3.0e6 : 0.0);
    if (cost < best_cost) { best_cost = cost; best = WidePlan{cfgs[i], parts}; }
  }
  return best;
}

}  // namespace

// Tile choice.  0: 256x256, 1: 256x64 (also: few-tile problems), 2: 32x256, 3: 288x192 (4: 256x128, the wide-decode planner's and by request only).  Measured on MI355X (in-process A/B,
// tools/bench_ops.py gemmcfg): 256x256 wins whenever the grid spans several rounds of the 256 CUs; the
// 288x192 tile wins where 256x256 leaves a single ragged round (N = 3072 at M = 4289: 204 tiles, but 240
// tiles of the smaller shape) and K is long enough to amortise its prologue.
int td_gemm_config_id(int M, int N, int K) {
  if (N <= 64) return 1;
  if (M <= 32) return 2;
  const long long t0 = (long long)((M + 255) / 256) * ((N + 255) / 256);
  const long long t3 = (long long)((M + 287) / 288) * ((N + 191) / 192);
  if (t0 < 230 && t3 <= 256 && t3 > t0 && K >= 6144 && M > 1024) return 3;
  // few 256x256 tiles (encoders, towers, short prefills: M of 77-1024 rows, N of a few thousand): a quarter-width tile
  // puts 4x the workgroups on the chip; measured 25-30 % faster below ~96 tiles, slower above (tools/bench_ops.py gemmsmall)
  return t0 < 96 ? 1 : 0;
}

int td_gemm_launch(const TdGemmParams& p, hipStream_t stream) {
  TD_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "td_gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  const int esz = (p.fp8 || p.i8) ? 1 : 2;
  TD_CHECK_ARG(!(p.fp8 && p.i8), "td_gemm: fp8 and int8 operands are exclusive");
  TD_CHECK_ARG(p.K % (128 / esz) == 0, "td_gemm: K=%d must be a multiple of %d", p.K, 128 / esz);
  TD_CHECK_ARG(p.N % 8 == 0, "td_gemm: N=%d must be a multiple of 8", p.N);
  TD_CHECK_ARG((p.conv_H > 0 || p.lda >= p.K) && p.ldc >= (p.C2 ? p.n_split : p.N), "td_gemm: bad leading dimensions");
  // 32-bit descriptor offsets: rows of the last tile may run up to one tile (<= 288 rows) past M / N before the range check drops them
  constexpr long long LIM = (1ll << 32) - (1ll << 16);
  TD_CHECK_ARG(((long long)(p.M + 288) * p.lda + p.K) * esz < LIM && (long long)(p.N + 288) * p.K * esz < LIM,
               "td_gemm: operand exceeds the 4 GiB buffer-descriptor range");
  TD_CHECK_ARG((long long)(p.M + 288) * p.ldc * (p.out_f32 ? 4 : 2) < LIM && (long long)(p.M + 288) * (p.C2 ? p.ldc2 : 0) * 2 < LIM &&
               (long long)(p.M + 288) * (p.res ? p.ldr : 0) * 2 < LIM && (long long)(p.g_M + 288) * p.ldc * 2 < LIM,
               "td_gemm: output / residual exceeds the 4 GiB buffer-descriptor range");
  TD_CHECK_ARG(((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.C) % 16 == 0 && p.lda % (16 / esz) == 0 && p.ldc % 8 == 0,
               "td_gemm: pointers / leading dimensions must be 16-byte aligned");
  if (p.res) TD_CHECK_ARG(p.ldr % 4 == 0, "td_gemm: ldr must be a multiple of 4");
  if (p.C2) TD_CHECK_ARG(p.ldc2 % 8 == 0 && p.n_split % 8 == 0 && p.n_split < p.N, "td_gemm: bad split-output arguments");
  if (p.g_M > 0) {
    TD_CHECK_ARG(p.g_A && p.g_W && p.g_C && !p.C2, "td_gemm: grouped launch needs A/W/C of the second problem and no split output");
    TD_CHECK_ARG(((uintptr_t)p.g_A | (uintptr_t)p.g_W | (uintptr_t)p.g_C) % 16 == 0, "td_gemm: grouped pointers must be 16-byte aligned");
  }
  if (p.conv_H > 0) {
    TD_CHECK_ARG(p.conv_Cin % 64 == 0 && p.K == 9 * p.conv_Cin && p.M == p.conv_H * p.conv_W && p.g_M == 0 && !p.C2,
                 "td_gemm(conv): need Cin %% 64 == 0, K == 9 Cin, M == H W (got Cin=%d K=%d M=%d H=%d W=%d)", p.conv_Cin, p.K, p.M, p.conv_H, p.conv_W);
    TD_CHECK_ARG(p.conv_up == 0 || (p.conv_H % 2 == 0 && p.conv_W % 2 == 0), "td_gemm(conv): upsampled output dims must be even");
    // output-channel tile: 64, 128 (the 128-channel 1024^2 / 512^2 layers of the VAE: a 256-wide tile would be half empty) or 256
    return p.N <= 64 ? launch_cfg<8, 1, true>(p, stream) : p.N <= 128 ? launch_cfg<8, 2, true>(p, stream) : launch_cfg<8, 4, true>(p, stream);
  }
  // M <= 64 without a tile override is a weight stream, not a tile problem (Qwen2-VL decode of up to 64 sequences, embedders, lm_head)
  if (p.glu_I) {
    TD_CHECK_ARG(p.M <= 64 && p.g_M == 0 && !p.fp8 && !p.i8 && !p.out_f32 && p.conv_H == 0, "td_gemm: the gated form exists for the skinny-M kernels only (M <= 64)");
    return td_gemv_launch(p, stream);
  }
  if (p.g_M == 0 && !p.fp8 && !p.i8 && !p.out_f32 && p.cfg < 0 && p.split_k == 0 && p.N % 4 == 0 && (p.M <= 16 || (p.M <= 64 && td_gemv_mfma_ok(p)))) return td_gemv_launch(p, stream);      // (a caller that sets split_k wants the tile kernels)
  int cfg = p.cfg >= 0 ? p.cfg : td_gemm_config_id(p.M + p.g_M, p.N, p.K * esz / 2);
  int parts = 1;
  if (p.split_k == -1 && p.cfg < 0 && p.M > 16 && p.M <= 256 && !p.fp8 && !p.i8 && p.g_M == 0 && !p.out_f32 && !p.q8) {
    const WidePlan w = plan_wide(p);
    cfg = w.cfg; parts = w.parts;
  } else {
    parts = plan_split_k(p, cfg);
  }
  const bool fuse_norm = p.sk_norm_w != nullptr;
  if (fuse_norm) {
    TD_CHECK_ARG(p.split_k == -1 && p.sk_norm_out && !p.C2 && p.N % 512 == 0 && p.N <= 4096 && p.sk_norm_ld % 8 == 0 && plan_split_k(p, cfg) >= 1 && !p.fp8 && !p.i8 &&
                     p.conv_H == 0 && p.g_M == 0 && !p.out_f32 && p.act == TD_ACT_NONE && !p.gate && !p.q8 && p.M > 16 &&
                     (long long)p.M * p.N * 4 <= (p.sk_ws ? p.sk_ws_bytes : SPLITK_POOL_BYTES),
                 "td_gemm(sk_norm): plain / bias / residual bf16 launches under split_k = -1, N %% 512 == 0, N <= 4096, M > 16");
  }
  if (parts > 1 || fuse_norm) {
    float* ws = p.sk_ws;
    if (!ws) {
      if (int rc = gemm_splitk_pool(stream, &ws)) return rc;
    }
    TdGemmParams q = p;
    q.split_k = 0; q.k_parts = parts; q.K = p.K / parts; q.ldw = p.K;
    q.C = (bf16_t*)ws; q.ldc = p.N; q.out_f32 = 1; q.bias = nullptr; q.res = nullptr; q.C2 = nullptr; q.n_split = 0;
    int rc;
    switch (cfg) {
      case 1: rc = launch_cfg<8, 1>(q, stream); break;
      case 2: rc = launch_cfg<1, 4>(q, stream); break;
      case 3: rc = launch_cfg<9, 3>(q, stream); break;
      case 4: rc = launch_cfg<8, 2>(q, stream); break;
      default: rc = launch_cfg<8, 4>(q, stream);
    }
    if (rc) return rc;
    if (fuse_norm) {
      const dim3 g(p.M);
      switch (p.N / 512) {
#define TD_CASE(n) case n: hipLaunchKernelGGL(td_gemm_splitk_reduce_norm_kernel<n>, g, dim3(n * 64), 0, stream, ws, parts, p.M, p.bias, p.res, p.ldr, p.C, p.ldc, p.sk_norm_w, p.sk_norm_out, p.sk_norm_ld, p.sk_norm_eps); break;
        TD_CASE(1) TD_CASE(2) TD_CASE(3) TD_CASE(4) TD_CASE(5) TD_CASE(6) TD_CASE(7) TD_CASE(8)
#undef TD_CASE
      }
      TD_CHECK_LAUNCH();
      return 0;
    }
    TD_GRID_1D(nblk, (long long)p.M * (p.N / 4), 256, "td_gemm(split-K reduce)");
    hipLaunchKernelGGL(td_gemm_splitk_reduce_kernel, dim3(nblk), dim3(256), 0, stream, ws, parts, p.M, p.N, p.bias, p.res, p.ldr, p.C, p.ldc, p.C2, p.ldc2, p.n_split);
    TD_CHECK_LAUNCH();
    return 0;
  }
  TD_CHECK_ARG(p.split_k <= 1 || parts > 1, "td_gemm: split_k=%d is not available for this problem (bf16 plain / bias / residual forms whose k-tile count it divides)", p.split_k);
  if (p.q8) {
    TD_CHECK_ARG(p.i8 && p.q8_inv && p.q8_amax && p.ldq8 % 16 == 0 && ((uintptr_t)p.q8) % 16 == 0 && (p.g_M == 0 || (p.g_q8 && p.g_q8_inv && p.g_q8_amax)),
                 "td_gemm(q8 output): int8 kernels only; needs the per-row inverse scales, the amax accumulators and 16-byte aligned rows");
    TD_CHECK_ARG((p.C2 ? (p.N - p.n_split) % 16 == 0 && p.n_split % 16 == 0 : p.N % 16 == 0) && (long long)(p.M + p.g_M + 288) * p.ldq8 < (1ll << 32) - (1ll << 16),
                 "td_gemm(q8 output): output widths must be multiples of 16");
    TD_CHECK_ARG(cfg != 3, "td_gemm(q8 output): the 288x192 tile owns 12 columns per lane; use tile 0 or 2");
  }
  if (p.i8) {
    TD_CHECK_ARG(p.a_scale && p.w_scale && (p.g_M == 0 || (p.g_a_scale && p.g_w_scale)) && p.conv_H == 0 && !p.out_f32,
                 "td_gemm(int8): row / column dequantisation scales are required; no conv / fp32-out form");
    switch (cfg) {
      case 2: return launch_cfg<1, 4, false, false, true>(p, stream);
      case 3: return launch_cfg<9, 3, false, false, true>(p, stream);
      default: return launch_cfg<8, 4, false, false, true>(p, stream);
    }
  }
  if (p.fp8) {
    TD_CHECK_ARG(p.a_scale && p.w_scale && (p.g_M == 0 || (p.g_a_scale && p.g_w_scale)) && p.conv_H == 0 && !p.out_f32,
                 "td_gemm(fp8): row / column dequantisation scales are required; no conv / fp32-out form");
    switch (cfg) {
      case 2: return launch_cfg<1, 4, false, true>(p, stream);
      case 3: return launch_cfg<9, 3, false, true>(p, stream);
      default: return launch_cfg<8, 4, false, true>(p, stream);
    }
  }
  switch (cfg) {
    case 1: return launch_cfg<8, 1>(p, stream);
    case 2: return launch_cfg<1, 4>(p, stream);
    case 3: return launch_cfg<9, 3>(p, stream);
    case 4: return launch_cfg<8, 2>(p, stream);
    default: return launch_cfg<8, 4>(p, stream);
  }
}
