// Skinny-M Linear for gfx950: y[M<=64, N] = epilogue(x[M,K] . W[N,K]^T), the weight-streaming case of the hot path --
// KV-cached Qwen2-VL decode (one token against 7.6 B parameters: SURVEY.md 8a row A7, HBM roofline), the pooled / timestep
// embedders of FLUX, lm_head.  The MFMA tile kernel launches N/256 workgroups here (14 for down_proj) and leaves the
// chip idle; this kernel is a plain HBM stream:
//   * a workgroup (4 waves) owns 4 consecutive output columns = 4 weight rows; its 256 threads stride the rows' 16-byte
//     chunks, so every wave-level load is a fully coalesced 1 KiB line run; loads are non-temporal (weights are read once);
//   * two chunk-columns are in flight per thread (8 weight loads + the matching x chunks, which hit L2);
//   * bf16 pairs go straight into v_dot2c_f32_bf16 (fp32 accumulate, no unpacking);
//   * wave shuffle + LDS cross-wave reduction, then the same epilogue semantics and bf16 rounding points as the tile kernel
//     (bias, activation, gate, residual, split output).
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "td_common.h"
#include "td_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ float dot8(const u32x4_t& a, const u32x4_t& b, float acc) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // copy the elements out first: __builtin_bit_cast applied directly to a vector-element lvalue reads element 0 (hipcc 7.2)
    const unsigned ua = a[q], ub = b[q];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ua), __builtin_bit_cast(bf16x2_t, ub), acc, false);
  }
  return acc;
}

__device__ __forceinline__ float act_rt(int act, float x) {
  switch (act) {
    case TD_ACT_GELU_TANH: return gelu_tanh_f(x);
    case TD_ACT_GELU_ERF: return gelu_erf_f(x);
    case TD_ACT_SILU: return silu_f(x);
    case TD_ACT_QUICK_GELU: return quick_gelu_f(x);
    default: return x;
  }
}

constexpr int R = 4;          // weight rows (output columns) per workgroup
constexpr int THREADS = 256;

template <int MR>
__global__ __launch_bounds__(THREADS) void td_gemv_bf16_kernel(const TdGemmParams p) {
  __shared__ float red[THREADS / 64][R][MR];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n0 = blockIdx.x * R;
  const int nchunk = p.K >> 3;
  const u32x4_t* wp[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    // gated mode: rows 0,1 = gate rows 2b, 2b+1; rows 2,3 = the matching up rows (glu_I further down the matrix)
    const int row = p.glu_I ? (blockIdx.x * 2 + (r & 1) + (r >> 1) * p.glu_I) : min(n0 + r, p.N - 1);
    wp[r] = (const u32x4_t*)(p.W + (size_t)row * p.K);
  }
  const u32x4_t* xp[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) xp[m] = (const u32x4_t*)(p.A + (size_t)min(m, p.M - 1) * p.lda);

  float acc[R][MR];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[r][m] = 0.f;

  int c = tid;
  for (; c + THREADS < nchunk; c += 2 * THREADS) {
    u32x4_t w0[R], w1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      w0[r] = __builtin_nontemporal_load(wp[r] + c);
      w1[r] = __builtin_nontemporal_load(wp[r] + c + THREADS);
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const u32x4_t x0 = xp[m][c], x1 = xp[m][c + THREADS];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r][m] = dot8(w1[r], x1, dot8(w0[r], x0, acc[r][m]));
    }
  }
  if (c < nchunk) {
    u32x4_t w0[R];
#pragma unroll
    for (int r = 0; r < R; ++r) w0[r] = __builtin_nontemporal_load(wp[r] + c);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const u32x4_t x0 = xp[m][c];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r][m] = dot8(w0[r], x0, acc[r][m]);
    }
  }

#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const float v = wave_sum(acc[r][m]);
      if (lane == 0) red[wid][r][m] = v;
    }
  __syncthreads();
  if (p.glu_I) {
    if (tid >= 2 * MR) return;
    const int i = tid / MR, m = tid % MR;
    if (m >= p.M) return;
    float gsum = 0.f, usum = 0.f;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) { gsum += red[w][i][m]; usum += red[w][i + 2][m]; }
    p.C[(size_t)m * p.ldc + blockIdx.x * 2 + i] = f2bf(rbf(silu_f(rbf(gsum))) * rbf(usum));
    return;
  }
  if (tid >= R * MR) return;
  const int r = tid / MR, m = tid % MR;
  const int n = n0 + r;
  if (n >= p.N || m >= p.M) return;
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) v += red[w][r][m];
  // epilogue: the tile kernel's semantics and rounding points (Linear output, activation, gate, residual each round)
  const bool second = p.C2 != nullptr && n >= p.n_split;
  const int act = second ? p.act2 : p.act;
  if (p.bias) v += bf2f(p.bias[n]);
  if (act != TD_ACT_NONE) v = act_rt(act, rbf(v));
  else {
    if (p.gate) v = rbf(v) * bf2f(p.gate[n]);
    if (p.res) v = rbf(v) + bf2f(p.res[(size_t)m * p.ldr + n]);
  }
  if (second) p.C2[(size_t)m * p.ldc2 + (n - p.n_split)] = f2bf(v);
  else p.C[(size_t)m * p.ldc + n] = f2bf(v);
}

// ---- 4 < M <= 64: the same weight stream on the matrix core ---------------------------------------------------------
// With more activation rows the dot-product form re-reads x from L2 MR times per weight chunk (at M = 16 x traffic is 8x the
// weight traffic and the kernel stops scaling).  Here a workgroup owns NR blocks of 16 weight rows, its 8 waves split K in
// 64-element steps, and every step is two v_mfma_f32_16x16x32_bf16 per (weight block, activation block) with the weight rows as
// the A operand and 16 activation rows as the B operand -- both fragments are loaded straight from global memory in MFMA operand
// layout (lane = row, 16 B of k), so there is no VALU work in the loop.  MB = ceil(M / 16) activation blocks share every weight
// fragment (batched decode of up to 64 sequences: one pass over the weights); x comes from L2, MB / NR bytes per weight byte.
// Accumulators are reduced across waves through LDS, one weight block at a time.
constexpr int MW = 8;   // waves per workgroup (K split)

// Few weight blocks (N = hidden: 96 blocks for the 2B decoder, 224 for the 7B one) leave most CUs without a workgroup.  Then K is
// also split across KS workgroups per block column (gridDim.y): each leaves its fp32 partial tiles in a workspace and draws a
// ticket; the LAST to arrive adds the KS partials in index order (the result does not depend on who finishes) and runs the epilogue.
// Hand-off as in the stream-K attention kernel: write-through (sc1) 16-byte stores, every storing wave drains, workgroup barrier,
// one agent-scope atomic; reader: one agent-scope acquire, barrier, plain loads.  The finisher re-arms the counter.
constexpr int SPLITK_MAX_WGS = 512;                       // block columns a split launch may have (sizes the counter array)
constexpr size_t SPLITK_HEADER_BYTES = SPLITK_MAX_WGS * 4;
constexpr size_t SPLITK_PART_BYTES = (size_t)SPLITK_MAX_WGS * 4 * 8 * 1024;   // block columns x MB x KS x (64 lanes x 16 B)

// LDS bytes of one workgroup: per wave a private slab of 2 NR weight + 2 XB activation 1-KiB fragments (one k-step; more than two
// activation blocks take turns in the slab, which keeps 2-3 workgroups per CU at 64 sequences), reused for the cross-wave
// reduction (MB KiB per wave) after the loop
template <int MB> constexpr int gemv_xb() { return MB > 2 ? 2 : MB; }     // activation blocks passing through the slab together
template <int MB, int NR> constexpr int gemv_lds_bytes() { return MW * (2 * NR + 2 * gemv_xb<MB>()) * 1024; }

template <int MB, int NR, bool DEEP>
__global__ __launch_bounds__(MW * 64) void td_gemv_mfma_kernel(const TdGemmParams p, char* ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XB = gemv_xb<MB>();
  constexpr int SLAB = (2 * NR + 2 * XB) * 1024;
  float (*red)[SLAB / 4] = (float (*)[SLAB / 4])smem;        // red[wave][(mb * 64 + lane) * 4 + i], MB KiB of each wave's slab
  __shared__ unsigned ticket_lds;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int nblk = p.glu_I ? p.glu_I / 8 : p.N / 16;         // 16-row weight blocks of the problem
  const int KS = gridDim.y, ky = blockIdx.y;
  // Operands through buffer descriptors: a k-step past this workgroup's range is sent beyond the descriptor and reads zeros, so
  // the loop body is branch-free.  A load instruction covers 8 rows x one whole 128-byte line each (lane = row l / 8, 16-byte chunk
  // l % 8): in MFMA operand layout (4 lanes per row) it would touch 16 half lines, and the other halves, fetched by the next
  // instruction, no longer find their lines in the 16 KiB L1 -- measured with a timing-only build: -17 % (16 sequences) to -27 %
  // (64) per decode step.  The fragments then pass through a wave-private LDS slab into operand layout: ds_write_b128 at lane x 16
  // (conflict-free), ds_read_b128 of row r, chunk 4 h + g; the chunk a lane FETCHES is XOR-ed with its row so those reads spread
  // over the banks.  LDS operations of a wave execute in order, so the slab needs no barrier between steps.
  // DEEP (long K: at least 4 steps per wave): all loads of SF steps -- up to 32 per lane -- are in flight before the first
  // LDS pass (a workgroup streams only 16 rows x K, its lifetime is a handful of memory round trips); otherwise hipcc threads
  // the loads between the steps at low register count, which keeps several workgroups per CU for short K.
  const unsigned w_rows = p.glu_I ? 2u * (unsigned)p.glu_I : (unsigned)p.N;
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)w_rows * p.K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)((((size_t)p.M - 1) * p.lda + p.K) * 2), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  const int lrow = lane >> 3;                                  // row of the 8-row half this lane fetches
  const int lchunk = (lane & 7) ^ (lrow & 7);                  // ... and which 16-byte chunk of the row's 128-byte line
  unsigned woff[NR][2], xoff[MB][2];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int bid = min((int)blockIdx.x * NR + nr, nblk - 1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // gated mode: rows 0-7 = gate rows 8b..8b+7, rows 8-15 = the matching up rows
      const int wr_ = p.glu_I ? (bid * 8 + lrow + h * p.glu_I) : bid * 16 + 8 * h + lrow;
      woff[nr][h] = (unsigned)(((size_t)wr_ * p.K + 8 * lchunk) * 2);
    }
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int h = 0; h < 2; ++h) xoff[mb][h] = (unsigned)(((size_t)min(16 * mb + 8 * h + lrow, p.M - 1) * p.lda + 8 * lchunk) * 2);
  char* slab = smem + wid * SLAB;
  u32x4_t* wr_ptr = (u32x4_t*)(slab + lane * 16);                                 // + fragment * 1024
  const char* rd_ptr[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) rd_ptr[h] = slab + r * 128 + (((4 * h + g) ^ (r & 7)) << 4);   // + block * 2048
  const int nk_all = p.K >> 6;                  // 64-element steps
  const int s_beg = (int)((long long)nk_all * ky / KS), nk = (int)((long long)nk_all * (ky + 1) / KS);   // this workgroup's steps
  f32x4_t acc[NR][MB];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[nr][mb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  constexpr int SF = DEEP ? 32 / (2 * NR + 2 * MB) : (MB * NR == 1 ? 2 : 1);    // k-steps per wave and iteration
  for (int s = s_beg + wid; s < nk; s += SF * MW) {
    u32x4_t w[SF][NR][2], x[SF][MB][2];
#pragma unroll
    for (int f = 0; f < SF; ++f) {
      const int st = s + f * MW;
      const bool ok = st < nk;
      const unsigned k0 = (unsigned)st * 128u;
#pragma unroll
      for (int nr = 0; nr < NR; ++nr)
#pragma unroll
        for (int h = 0; h < 2; ++h) w[f][nr][h] = __builtin_amdgcn_raw_buffer_load_b128(rsW, ok ? woff[nr][h] + k0 : OOB, 0, 0);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int h = 0; h < 2; ++h) x[f][mb][h] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? xoff[mb][h] + k0 : OOB, 0, 0);
    }
    if (DEEP) __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise threads the loads between the steps and keeps 4-6 of them in flight
#pragma unroll
    for (int f = 0; f < SF; ++f) {
      // block b of the slab: fragments 2b (rows 0-7) and 2b + 1 (rows 8-15); weight blocks first, then activation blocks
#pragma unroll
      for (int nr = 0; nr < NR; ++nr)
#pragma unroll
        for (int h = 0; h < 2; ++h) wr_ptr[(2 * nr + h) * 64] = w[f][nr][h];
      bf16x8_t wf[NR][2];
#pragma unroll
      for (int m0 = 0; m0 < MB; m0 += XB) {       // XB activation blocks at a time through the slab
#pragma unroll
        for (int j = 0; j < XB; ++j)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if (m0 + j < MB) wr_ptr[(2 * (NR + j) + h) * 64] = x[f][m0 + j < MB ? m0 + j : 0][h];
        bf16x8_t xf[XB][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (m0 == 0) {
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) wf[nr][h] = *(const bf16x8_t*)(rd_ptr[h] + nr * 2048);
          }
#pragma unroll
          for (int j = 0; j < XB; ++j) xf[j][h] = *(const bf16x8_t*)(rd_ptr[h] + (NR + j) * 2048);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int j = 0; j < XB; ++j)
              if (m0 + j < MB) acc[nr][m0 + j < MB ? m0 + j : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nr][h], xf[j][h], acc[nr][m0 + j < MB ? m0 + j : 0], 0, 0, 0);
      }
    }
  }
  __syncthreads();          // every wave is done with its slab before the reduction reuses the space

  // wave mb (< MB) finishes activation block mb: v = fp32 tile column (lane & 15) = activation row, rows 4 (lane >> 4) + i = weight rows
  auto epilogue = [&](int bid, float (&v)[4]) {
    const int m = 16 * wid + r;
    if (p.glu_I) {   // lanes g = 0,1 hold gate rows 4g+i, their partners 32 lanes up the matching up rows
      float u[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) u[i] = __shfl(v[i], (lane + 32) & 63, 64);
      if (g < 2 && m < p.M) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = rbf(silu_f(rbf(v[i]))) * rbf(u[i]);
        *(u32x2_t*)(p.C + (size_t)m * p.ldc + bid * 8 + 4 * g) = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      }
      return;
    }
    const int n = bid * 16 + 4 * g;
    if (m >= p.M) return;
    const bool second = p.C2 != nullptr && n >= p.n_split;
    const int act = second ? p.act2 : p.act;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float y = v[i];
      if (p.bias) y += bf2f(p.bias[n + i]);
      if (act != TD_ACT_NONE) y = act_rt(act, rbf(y));
      else {
        if (p.gate) y = rbf(y) * bf2f(p.gate[n + i]);
        if (p.res) y = rbf(y) + bf2f(p.res[(size_t)m * p.ldr + n + i]);
      }
      v[i] = y;
    }
    const u32x2_t o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    if (second) *(u32x2_t*)(p.C2 + (size_t)m * p.ldc2 + (n - p.n_split)) = o;
    else *(u32x2_t*)(p.C + (size_t)m * p.ldc + n) = o;
  };
  // partial tile of (block column, nr, mb, k-part): 64 lanes x 16 bytes
  const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + SPLITK_HEADER_BYTES), 0, (unsigned)SPLITK_PART_BYTES, 0x00020000);
  auto part = [&](int nr, int mb, int part_k) -> unsigned {
    return (unsigned)((((blockIdx.x * NR + nr) * MB + mb) * KS + part_k) * 64 + lane) * 16u;
  };

#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    if (nr) __syncthreads();
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      *(f32x4_t*)&red[wid][(mb * 64 + lane) * 4] = acc[nr][mb];
    __syncthreads();
    const int bid = blockIdx.x * NR + nr;
    if (wid >= MB || bid >= nblk) continue;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < MW; ++w) {
      const f32x4_t t = *(const f32x4_t*)&red[w][(wid * 64 + lane) * 4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += t[i];
    }
    if (KS == 1) epilogue(bid, v);
    else __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{as_u32(v[0]), as_u32(v[1]), as_u32(v[2]), as_u32(v[3])}, rsP, part(nr, wid, ky), 0, 16);   // aux 16 = sc1
  }
  if (KS == 1) return;
  // ---- split K: publish, draw the ticket, the last arriver sums the parts in index order and finishes ------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned* cnt = (unsigned*)ws + blockIdx.x;
  if (tid == 0) ticket_lds = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket_lds != (unsigned)(KS - 1)) return;
  if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every ticket drawn: ready for the next launch
  if (wid >= MB) return;
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int bid = blockIdx.x * NR + nr;
    if (bid >= nblk) continue;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < KS; ++k) {
      const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(rsP, part(nr, wid, k), 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += as_f32(t[i]);
    }
    epilogue(bid, v);
  }
}

// one hand-off workspace per (device, stream): launches on one stream never overlap
int splitk_workspace(hipStream_t stream, char** out) {
  struct Entry { int dev; hipStream_t stream; char* ws; };
  static std::mutex mu;
  static std::vector<Entry> pool;
  int dev = 0;
  TD_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  for (auto& e : pool)
    if (e.dev == dev && e.stream == stream) { *out = e.ws; return 0; }
  char* w = nullptr;
  TD_CHECK_HIP(hipMalloc((void**)&w, SPLITK_HEADER_BYTES + SPLITK_PART_BYTES));
  TD_CHECK_HIP(hipMemsetAsync(w, 0, SPLITK_HEADER_BYTES, stream));      // on the launching stream (ordered before the kernel)
  pool.push_back(Entry{dev, stream, w});
  *out = w;
  return 0;
}

template <int MB, int NR, bool DEEP>
int launch_one(const TdGemmParams& p, dim3 grid, char* ws, hipStream_t stream) {
  constexpr int lds = gemv_lds_bytes<MB, NR>();
  if (lds + 64 >= 64 * 1024) {  // dynamic + the static ticket word reach the default 64 KiB limit: raise it once per device
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    TD_CHECK_HIP(hipGetDevice(&dev));
    if (!((done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
      TD_CHECK_HIP(hipFuncSetAttribute((const void*)td_gemv_mfma_kernel<MB, NR, DEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
  }
  hipLaunchKernelGGL((td_gemv_mfma_kernel<MB, NR, DEEP>), grid, dim3(MW * 64), lds, stream, p, ws);
  return 0;
}

template <int MB>
int launch_mfma(const TdGemmParams& p, hipStream_t stream) {
  const int nblk = p.glu_I ? p.glu_I / 8 : p.N / 16;
  // two weight blocks per workgroup halve the x re-reads from L2; only where the grid still oversubscribes the chip
  if (MB > 1 && nblk >= 1024) {
    const dim3 grid((nblk + 1) / 2);
    return (p.K >> 6) >= 4 * MW ? launch_one<MB, 2, true>(p, grid, nullptr, stream) : launch_one<MB, 2, false>(p, grid, nullptr, stream);
  }
  // few block columns: split K over workgroups too, as long as every wave of a workgroup keeps at least one 64-element step
  int ks = 1;
  if (nblk < SPLITK_MAX_WGS / 2) ks = std::max(1, std::min({8, SPLITK_MAX_WGS / nblk, (p.K >> 6) / MW}));
  char* ws = nullptr;
  if (ks > 1) {
    if (int rc = splitk_workspace(stream, &ws)) return rc;
  }
  const dim3 grid(nblk, ks);
  return (p.K >> 6) / ks >= 4 * MW ? launch_one<MB, 1, true>(p, grid, ws, stream) : launch_one<MB, 1, false>(p, grid, ws, stream);
}

int launch_mfma_rows(const TdGemmParams& p, hipStream_t stream) {
  if (p.M <= 16) return launch_mfma<1>(p, stream);
  if (p.M <= 32) return launch_mfma<2>(p, stream);
  if (p.M <= 48) return launch_mfma<3>(p, stream);
  return launch_mfma<4>(p, stream);
}

}  // namespace

// shapes the matrix-core weight stream takes (td_gemm_launch asks before routing 16 < M <= 64 here)
bool td_gemv_mfma_ok(const TdGemmParams& p) {
  const long long w_rows = p.glu_I ? 2ll * p.glu_I : p.N;          // operands sit behind 32-bit buffer descriptors
  if (w_rows * p.K * 2 >= 0xFFFFFF00ll || ((long long)(p.M - 1) * p.lda + p.K) * 2 >= 0xFFFFFF00ll) return false;
  return p.K % 64 == 0 && p.N % 16 == 0 && p.ldc % 4 == 0 && p.lda % 8 == 0 && (!p.C2 || (p.ldc2 % 4 == 0 && p.n_split % 4 == 0));
}


int td_gemv_launch(const TdGemmParams& p, hipStream_t stream) {
  TD_CHECK_ARG(p.M >= 1 && p.M <= 64 && p.N % R == 0 && p.K % 8 == 0 && p.lda % 8 == 0, "td_gemv: needs M <= 64, N %% 4 == 0, K %% 8 == 0");
  TD_CHECK_ARG(((uintptr_t)p.A | (uintptr_t)p.W) % 16 == 0, "td_gemv: operands must be 16-byte aligned");
  if (p.glu_I) {
    TD_CHECK_ARG(p.N == p.glu_I && p.glu_I % 8 == 0 && !p.bias && !p.gate && !p.res && !p.C2 && p.act == TD_ACT_NONE && p.ldc % 4 == 0,
                 "td_gemv(glu): N must equal glu_I (multiple of 8), no bias / gate / residual / split");
    const bool mfma = p.K % 64 == 0 && 2ll * p.glu_I * p.K * 2 < 0xFFFFFF00ll && ((long long)(p.M - 1) * p.lda + p.K) * 2 < 0xFFFFFF00ll;
    TD_CHECK_ARG(p.M <= 16 || mfma, "td_gemv(glu): more than 16 rows need K %% 64 == 0 and operands under 4 GiB");
    if (p.M > 4 && mfma) {
      if (int rc = launch_mfma_rows(p, stream)) return rc;
    } else if (p.M == 1) hipLaunchKernelGGL(td_gemv_bf16_kernel<1>, dim3(p.glu_I / 2), dim3(THREADS), 0, stream, p);
    else if (p.M == 2) hipLaunchKernelGGL(td_gemv_bf16_kernel<2>, dim3(p.glu_I / 2), dim3(THREADS), 0, stream, p);
    else if (p.M <= 4) hipLaunchKernelGGL(td_gemv_bf16_kernel<4>, dim3(p.glu_I / 2), dim3(THREADS), 0, stream, p);
    else if (p.M <= 8) hipLaunchKernelGGL(td_gemv_bf16_kernel<8>, dim3(p.glu_I / 2), dim3(THREADS), 0, stream, p);
    else hipLaunchKernelGGL(td_gemv_bf16_kernel<16>, dim3(p.glu_I / 2), dim3(THREADS), 0, stream, p);
    TD_CHECK_LAUNCH();
    return 0;
  }
  if (p.M > 4 && td_gemv_mfma_ok(p)) {
    if (int rc = launch_mfma_rows(p, stream)) return rc;
    TD_CHECK_LAUNCH();
    return 0;
  }
  TD_CHECK_ARG(p.M <= 16, "td_gemv: more than 16 rows need the matrix-core form (K %% 64, N %% 16, ldc %% 4 == 0)");
  const dim3 grid(p.N / R), block(THREADS);
  if (p.M == 1) hipLaunchKernelGGL(td_gemv_bf16_kernel<1>, grid, block, 0, stream, p);
  else if (p.M == 2) hipLaunchKernelGGL(td_gemv_bf16_kernel<2>, grid, block, 0, stream, p);
  else if (p.M <= 4) hipLaunchKernelGGL(td_gemv_bf16_kernel<4>, grid, block, 0, stream, p);
  else if (p.M <= 8) hipLaunchKernelGGL(td_gemv_bf16_kernel<8>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(td_gemv_bf16_kernel<16>, grid, block, 0, stream, p);
  TD_CHECK_LAUNCH();
  return 0;
}
