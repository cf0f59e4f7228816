// Probe: exact lane/element mapping of ds_read_b64_tr_b16 on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((address_space(3))) short4v lds_short4;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  // lane l supplies the address of elements 4l..4l+3
  short4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(lds + 4 * threadIdx.x));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}
int main() {
  short* d;
  hipMalloc(&d, 256 * sizeof(short));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (L%2d,e%d)", h[l * 4 + j] / 4, h[l * 4 + j] % 4);
    printf("\n");
  }
  return 0;
}
