// Checks the LDS-free wave reductions of csrc/common.h (DPP + v_permlane*_swap) against a host sum.
// build: hipcc -O3 --offload-arch=gfx950 -I torch-mednet_amd/csrc tools/probes/wave_sum_probe.hip -o tools/probes/wave_sum_probe
#include "common.h"
#include <vector>
using namespace mednet;
__global__ void k(const float* in, float* out_f, double* out_d, float* cls) {
  const int l = threadIdx.x;
  out_f[l] = wave_sum(in[l]);
  out_d[l] = wave_sum((double)in[l] * 1.0000001);
  cls[0 * 64 + l] = lane_class_sum<2>(in[l]);
  cls[1 * 64 + l] = lane_class_sum<4>(in[l]);
  cls[2 * 64 + l] = lane_class_sum<8>(in[l]);
  cls[3 * 64 + l] = lane_class_sum<16>(in[l]);
  cls[4 * 64 + l] = lane_class_sum<32>(in[l]);
}
int main() {
  std::vector<float> h(64);
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37 % 64) + 1) + 0.25f * (i % 3);
  float *in, *of, *cls;
  double* od;
  hipMalloc(&in, 256); hipMalloc(&of, 256); hipMalloc(&od, 512); hipMalloc(&cls, 5 * 256);
  hipMemcpy(in, h.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, of, od, cls);
  std::vector<float> rf(64), rc(5 * 64);
  std::vector<double> rd(64);
  hipMemcpy(rf.data(), of, 256, hipMemcpyDeviceToHost);
  hipMemcpy(rd.data(), od, 512, hipMemcpyDeviceToHost);
  hipMemcpy(rc.data(), cls, 5 * 256, hipMemcpyDeviceToHost);
  double tot = 0;
  for (float v : h) tot += v;
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    if (fabs(rf[l] - tot) > 1e-3) { if (bad++ < 4) printf("wave_sum(float) lane %d: %f vs %f\n", l, rf[l], tot); }
    if (fabs(rd[l] - tot * 1.0000001) > 1e-6) { if (bad++ < 8) printf("wave_sum(double) lane %d: %.9f vs %.9f\n", l, rd[l], tot * 1.0000001); }
  }
  const int strides[5] = {2, 4, 8, 16, 32};
  for (int s = 0; s < 5; ++s)
    for (int l = 0; l < 64; ++l) {
      double e = 0;
      for (int j = l % strides[s]; j < 64; j += strides[s]) e += h[j];
      if (fabs(rc[s * 64 + l] - e) > 1e-3) { if (bad++ < 400 && (l < 2 || l == 17 || l == 33 || l == 63)) printf("class_sum<%d> lane %d: %f vs %f\n", strides[s], l, rc[s * 64 + l], e); }
    }
  printf(bad ? "FAILED (%d mismatches)\n" : "wave reductions OK\n", bad);
  return bad != 0;
}
