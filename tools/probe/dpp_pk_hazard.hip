// Does a DPP read of a VGPR written by v_pk_add_f32 two wait states earlier (what hipcc emits for the cross-lane sums of
// csrc/layernorm.hip: v_pk_add_f32 ; s_nop 1 ; v_mov_b32_dpp) always see the new value on gfx950?
// Every wave runs the exact instruction sequence of the compiled group_sum (inline asm, fixed registers) next to the same
// sequence with s_nop 7 in front of every DPP and counts disagreements.  Configurations: one wave per SIMD (64-thread
// blocks, 40 KB of LDS each -> 4 per CU), 2 and 4 waves per SIMD, and each of them next to an LDS-DMA / barrier heavy
// kernel on a second stream.   hipcc --offload-arch=gfx950 -O3 -o dpp_pk_hazard.bin dpp_pk_hazard.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define DPPC "row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define CHAIN(NOP)                                                          \
  "v_mov_b32 v20, %2\n v_mov_b32 v21, %3\n s_nop 7\n"                       \
  "v_mov_b32_dpp v22, v20 quad_perm:[1,0,3,2] " DPPC "\n"                   \
  "v_mov_b32_dpp v23, v21 quad_perm:[1,0,3,2] " DPPC "\n"                   \
  "v_pk_add_f32 v[22:23], v[20:21], v[22:23]\n" NOP                         \
  "v_mov_b32_dpp v20, v22 quad_perm:[2,3,0,1] " DPPC "\n"                   \
  "v_mov_b32_dpp v21, v23 quad_perm:[2,3,0,1] " DPPC "\n"                   \
  "v_pk_add_f32 v[22:23], v[22:23], v[20:21]\n" NOP                         \
  "v_mov_b32_dpp v20, v22 row_half_mirror " DPPC "\n"                       \
  "v_mov_b32_dpp v21, v23 row_half_mirror " DPPC "\n"                       \
  "v_pk_add_f32 v[22:23], v[22:23], v[20:21]\n" NOP                         \
  "v_mov_b32_dpp v20, v22 row_mirror " DPPC "\n"                            \
  "v_mov_b32_dpp v21, v23 row_mirror " DPPC "\n"                            \
  "v_pk_add_f32 v[22:23], v[22:23], v[20:21]\n"                             \
  "s_nop 7\n v_mov_b32 %0, v22\n v_mov_b32 %1, v23\n"

// the tail of the compiled group_sum<64>: v_pk_add_f32 ; ds_bpermute_b32 x 2 (reading both halves of the packed result as DATA,
// no wait state in between) ; s_waitcnt lgkmcnt(0) ; v_pk_add_f32  -- twice (lane ^ 16, lane ^ 32)
#define BCHAIN(NOP)                                                         \
  "v_mov_b32 v20, %2\n v_mov_b32 v21, %3\n v_mov_b32 v24, %4\n v_mov_b32 v25, %5\n s_nop 7\n"   \
  "v_pk_add_f32 v[22:23], v[20:21], v[20:21]\n" NOP                         \
  "ds_bpermute_b32 v20, v24, v22\n"                                         \
  "ds_bpermute_b32 v21, v24, v23\n"                                         \
  "s_waitcnt lgkmcnt(0)\n"                                                  \
  "v_pk_add_f32 v[22:23], v[22:23], v[20:21]\n" NOP                         \
  "ds_bpermute_b32 v20, v25, v22\n"                                         \
  "ds_bpermute_b32 v21, v25, v23\n"                                         \
  "s_waitcnt lgkmcnt(0)\n"                                                  \
  "v_pk_add_f32 v[22:23], v[22:23], v[20:21]\n"                             \
  "s_nop 7\n v_mov_b32 %0, v22\n v_mov_b32 %1, v23\n"

__global__ void hazard_kernel(unsigned long long* bad, int iters, unsigned seed) {
  extern __shared__ float pad[];
  const int lane = threadIdx.x & 63;
  unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
  unsigned long long nb = 0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    float a = (float)(int)(s >> 8) * (1.f / 8388608.f) - 1.f;
    s = s * 1664525u + 1013904223u;
    float b = (float)(int)(s >> 8) * (1.f / 8388608.f) - 1.f;
    float f0, f1, g0, g1;
    asm volatile(CHAIN("s_nop 1\n") : "=v"(f0), "=v"(f1) : "v"(a), "v"(b) : "v20", "v21", "v22", "v23");
    asm volatile(CHAIN("s_nop 7\n s_nop 7\n") : "=v"(g0), "=v"(g1) : "v"(a), "v"(b) : "v20", "v21", "v22", "v23");
    nb += (__float_as_uint(f0) != __float_as_uint(g0)) + (__float_as_uint(f1) != __float_as_uint(g1));
    const int a16 = (lane ^ 16) << 2, a32 = (lane ^ 32) << 2;
    asm volatile(BCHAIN("") : "=v"(f0), "=v"(f1) : "v"(a), "v"(b), "v"(a16), "v"(a32) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
    asm volatile(BCHAIN("s_nop 7\n") : "=v"(g0), "=v"(g1) : "v"(a), "v"(b), "v"(a16), "v"(a32) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
    nb += ((unsigned long long)((__float_as_uint(f0) != __float_as_uint(g0)) + (__float_as_uint(f1) != __float_as_uint(g1)))) << 32;
    if (lane == 0 && it == iters + 5) pad[0] = f0;
  }
  if (nb) atomicAdd(bad, nb);
}

// noise on the second stream: LDS-DMA + barriers + MFMA, like the grouped weight gradient
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(512) void noise_kernel(const float* src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* p = src + ((size_t)blockIdx.x * 512 + threadIdx.x) * 4;
  for (int it = 0; it < iters; ++it) {
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(p + (size_t)j * 1048576), (lds_void_t*)(sm + (wave * 4 + j) * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 a = *reinterpret_cast<const bf16x8*>(sm + ((lane * 16 + it * 1024) & 32767));
    for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc, 0, 0, 0);
    __builtin_amdgcn_s_barrier();
  }
  if (acc[0] == 123.f) out[threadIdx.x] = acc[1];
}

int main() {
  unsigned long long* bad;
  hipMalloc(&bad, 8);
  float* src; hipMalloc(&src, (size_t)64 << 20);
  hipMemset(src, 0, (size_t)64 << 20);
  float* out; hipMalloc(&out, 4096);
  hipStream_t s0, s1;
  hipStreamCreate(&s0); hipStreamCreate(&s1);
  hipFuncSetAttribute((const void*)hazard_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipFuncSetAttribute((const void*)noise_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  struct Cfg { const char* name; int threads, lds, grid; } cfgs[] = {
      {"1 wave / SIMD (64 threads, 40 KB LDS)", 64, 40 * 1024, 4096},
      {"1 wave / CU   (64 threads, 150 KB LDS)", 64, 150 * 1024, 1024},
      {"2 waves / SIMD (256 threads, 48 KB LDS)", 256, 48 * 1024, 2048},
      {"8 waves / SIMD (256 threads, no LDS)", 256, 0, 8192}};
  for (int noise = 0; noise < 2; ++noise)
    for (auto& c : cfgs) {
      hipMemset(bad, 0, 8);
      const int iters = 20000;
      for (int rep = 0; rep < 4; ++rep) {
        if (noise) hipLaunchKernelGGL(noise_kernel, dim3(512), dim3(512), 32768, s1, src, out, 400);
        hipLaunchKernelGGL(hazard_kernel, dim3(c.grid), dim3(c.threads), c.lds, s0, bad, iters, 1234u + rep);
      }
      hipDeviceSynchronize();
      unsigned long long h = 0;
      hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
      printf("%-44s %s: DPP chain %llu, bpermute chain %llu disagreements in %.3g reductions each\n", c.name,
             noise ? "+ LDS-DMA/MFMA kernel on a 2nd stream" : "alone", h & 0xffffffffull, h >> 32, 4.0 * iters * c.grid * c.threads * 2);
    }
  return 0;
}
