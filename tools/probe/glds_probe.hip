// Probe: where does global_load_lds_dwordx4 put each lane's 16 bytes?  GPU box only.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
__global__ void k(const short* g, short* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 1024; i += 64) ((short*)lds)[i] = -1;
  __syncthreads();
  const int lane = threadIdx.x;
  // lane l loads the 8 shorts g[l*8 .. l*8+7] (value = l*8 + e)
  __builtin_amdgcn_global_load_lds((gbl_void_t*)(g + lane * 8), (lds_void_t*)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = ((short*)lds)[i];
}
int main() {
  short h[1024], *dg, *dout;
  for (int i = 0; i < 512; ++i) h[i] = (short)i;
  hipMalloc(&dg, 1024); hipMalloc(&dout, 2048);
  hipMemcpy(dg, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, dg, dout);
  hipMemcpy(h, dout, 2048, hipMemcpyDeviceToHost);
  for (int i = 0; i < 1024; i += 8) { if (i >= 64 && i < 448 && (i % 128) >= 16) continue; printf("lds[%4d..]: %d %d %d %d %d %d %d %d\n", i, h[i],h[i+1],h[i+2],h[i+3],h[i+4],h[i+5],h[i+6],h[i+7]); }
  return 0;
}
