// Probe of the wgrad fragment read (ds_read_b64_tr_b16 on the swizzled [64][128] LDS tile); GPU box only.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ int wg_swz(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
__global__ void k(short* out, int tok0, int col0, const short* src) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {  // 4 waves, each DMAs its own 16 rows (4 pieces), exactly like wgrad_glds_kernel::issue
    const int prow = lane >> 4, pslot = lane & 15;
    unsigned char* sa = tile + wave * 16 * 256;
    for (int j = 0; j < 4; ++j) {
      const int r = wave * 16 + j * 4 + prow;
      const int q = pslot ^ wg_swz(r);
      const short* s = src + r * 128 + (q << 3);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)s, (lds_void_t*)(sa + j * 4 * 256), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (wave != 1) return;
  const int p = lane & 15, g = lane >> 4;
  const int n = col0 + ((p & 3) << 2);
  for (int half = 0; half < 2; ++half) {
    const int r = tok0 + g * 8 + half * 4 + (p >> 2);
    const unsigned char* a = tile + r * 256 + ((((n >> 3) ^ wg_swz(r))) << 4) + ((n & 7) << 1);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
    for (int j = 0; j < 4; ++j) out[lane * 8 + half * 4 + j] = v[j];
  }
}
int main() {
  short* d; short h[512]; short* dsrc; static short hs[64 * 128];
  for (int i = 0; i < 64 * 128; ++i) hs[i] = (short)i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&dsrc, sizeof(hs));
  hipMemcpy(dsrc, hs, sizeof(hs), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 65536, 0, d, 0, 16, dsrc);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) { if (l % 16 > 2 && l % 16 < 15) continue;
    printf("lane %2d (c %2d g %d):", l, l & 15, l >> 4);
    for (int s = 0; s < 8; ++s) printf(" (%d,%d)", h[l*8+s] / 128, h[l*8+s] % 128);
    printf("\n"); }
  return 0;
}
