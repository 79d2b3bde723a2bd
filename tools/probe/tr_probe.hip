// Probe of ds_read_b64_tr_b16 semantics on gfx950 (run on the GPU box): hipcc --offload-arch=gfx950 tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(short* out, const int* addr_elems) {
  __shared__ __attribute__((aligned(16))) short buf[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) buf[i] = (short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(buf + addr_elems[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  int h_addr[64]; short h_out[256];
  int* d_addr; short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int test = 0; test < 2; ++test) {
    for (int l = 0; l < 64; ++l) {
      int p = l & 15, g = l >> 4;
      if (test == 0) h_addr[l] = l * 4;                                   // lane-linear
      else h_addr[l] = g * 2000 + (p >> 2) * 100 + (p & 3) * 4;           // piece p = row p/4 (stride 100), col group p%4
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_addr);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("test %d\n", test);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %5d %5d %5d %5d\n", l, h_addr[l], h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
  }
  return 0;
}
