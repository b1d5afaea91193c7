// Where hipcc (clang 22, ROCm 7.2) waits for an LDS-DMA in flight:  hipcc -O3 --offload-arch=gfx950 --cuda-device-only -DMODE=n -S
//   MODE 0  plain load through an integer-cast LDS address      -> no vmcnt wait in front of the ds_read_b128
//   MODE 1  __builtin_amdgcn_ds_read_tr16_b64_v4i16             -> s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16
//   MODE 2  the same with the address laundered through asm      -> s_waitcnt vmcnt(0) all the same
//   MODE 3  ds_read_b64_tr_b16 as inline asm + a counted lgkmcnt -> no vmcnt wait (csrc/attention_common.h uses this form)
// Compile-only probe ("where the next tile's K | V is waited for").
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
#define TD_LDS __attribute__((address_space(3)))
extern __shared__ __attribute__((aligned(16))) char smem[];
__global__ void k(const char* g, unsigned* out, int n) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 1u << 20, 0x00020000);
  unsigned lds0 = (unsigned)(uintptr_t)(TD_LDS char*)smem;
  unsigned acc = 0;
  for (int i = 0; i < n; ++i) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (TD_LDS void*)(smem + 16384), 16, threadIdx.x * 16 + i * 1024, 0, 0, 0);
#if MODE == 0
    u32x4_t v = *(const TD_LDS u32x4_t*)(uintptr_t)(lds0 + threadIdx.x * 16);
    acc += v[0] + v[3];
#elif MODE == 1
    bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TD_LDS bf16x4_t*)(uintptr_t)(lds0 + threadIdx.x * 8));
    acc += v[0] + v[3];
#elif MODE == 2
    unsigned a = lds0 + threadIdx.x * 8;
    asm volatile("" : "+v"(a));
    bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((TD_LDS bf16x4_t*)(uintptr_t)a);
    acc += v[0] + v[3];
#elif MODE == 3
    unsigned a = lds0 + threadIdx.x * 8;
    bf16x4_t v, w;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(2048));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(w) : "v"(a), "n"(4096));
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(v), "+v"(w) : "n"(0));
    acc += v[0] + v[3] + w[1];
#endif
    __builtin_amdgcn_s_barrier();
  }
  out[threadIdx.x] = acc;
}
