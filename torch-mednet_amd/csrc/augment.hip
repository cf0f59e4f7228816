// On-device intensity augmentation of a batch of training patches: the three batchgenerators transforms the reference
// composes (examples/train_seg.py:82-86: additive brightness, gamma, contrast; applied per sample at dataset.py:340-341),
// behind the device patch sampler.  The random numbers are drawn on the host in batchgenerators' order
// (mednet_hip/sampler.py); what needs the data -- the sample's min / max for the gamma map, each channel's mean / min / max
// for the contrast step -- stays on the device, so a batch costs five launches and no synchronisation:
//   1. per (sample, channel): min, max of the crop                 (one read)
//   2. per sample: range after brightness; per channel: its ends after the gamma map
//   3. per (sample, channel): sum of the gamma-mapped values       (one read)
//   4. per (sample, channel): mean
//   5. out = clip((g(x + add) - mean) * factor + mean, lo, hi)     (one read, one write; in place)
// HBM-bound: 16 bytes per voxel (three reads + one write of fp32).
#include "common.h"

namespace mednet {

constexpr int AUG_BLOCK = 256 * 8;

struct AugStat {  // per (sample, channel), fp32 x 8
  float mn, mx;        // of the crop
  float lo, hi;        // after brightness + gamma (= range the contrast step preserves)
  float gmin, grange;  // sample-wide minimum and range after brightness (gamma map's normalisation)
  float mean, pad;
};

__device__ __forceinline__ float gamma_map(float x, float add, float gmin, float grange, float gamma) {
  // np.power(((x - minm) / float(rnge + 1e-7)), gamma) * rnge + minm on the brightened value
  const float t = (x + add - gmin) / (grange + 1e-7f);
  return powf(fmaxf(t, 0.f), gamma) * grange + gmin;
}

__global__ __launch_bounds__(256) void aug_minmax_kernel(const float* __restrict__ x, float* __restrict__ part, size_t spatial) {
  __shared__ float s0[4], s1[4];
  const float* p = x + (size_t)blockIdx.y * spatial;
  float mn = INFINITY, mx = -INFINITY;
  const size_t v0 = (size_t)blockIdx.x * AUG_BLOCK;
  for (int it = 0; it < 8; ++it) {
    const size_t v = v0 + (size_t)it * 256 + threadIdx.x;
    if (v < spatial) {
      const float t = p[v];
      mn = fminf(mn, t);
      mx = fmaxf(mx, t);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, off, 64));
    mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    s0[threadIdx.x >> 6] = mn;
    s1[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
    o[0] = fminf(fminf(s0[0], s0[1]), fminf(s0[2], s0[3]));
    o[1] = fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3]));
  }
}

// one 64-thread workgroup per sample: channel extrema, sample range after brightness, ends after the gamma map
__global__ __launch_bounds__(64) void aug_finalize1_kernel(const float* __restrict__ part, const float* __restrict__ params,
                                                           AugStat* __restrict__ st, int c, int nblocks) {
  const int b = blockIdx.x;
  __shared__ float cmn[64], cmx[64];
  for (int ch = 0; ch < c; ++ch) {
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nblocks; i += 64) {
      const float* o = part + (((size_t)b * c + ch) * nblocks + i) * 2;
      mn = fminf(mn, o[0]);
      mx = fmaxf(mx, o[1]);
    }
    for (int off = 32; off > 0; off >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, off, 64));
      mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    }
    if (threadIdx.x == 0 && ch < 64) {
      cmn[ch] = mn;
      cmx[ch] = mx;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float gmin = INFINITY, gmax = -INFINITY;
    for (int ch = 0; ch < c; ++ch) {
      const float add = params[((size_t)b * c + ch) * 3];
      gmin = fminf(gmin, cmn[ch] + add);
      gmax = fmaxf(gmax, cmx[ch] + add);
    }
    const float grange = gmax - gmin;
    for (int ch = 0; ch < c; ++ch) {
      const float* pr = params + ((size_t)b * c + ch) * 3;
      AugStat& s = st[(size_t)b * c + ch];
      s.mn = cmn[ch];
      s.mx = cmx[ch];
      s.gmin = gmin;
      s.grange = grange;
      s.lo = gamma_map(cmn[ch], pr[0], gmin, grange, pr[1]);  // the map is monotone: a channel's ends map to its ends
      s.hi = gamma_map(cmx[ch], pr[0], gmin, grange, pr[1]);
      s.mean = 0.f;
      s.pad = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void aug_sum_kernel(const float* __restrict__ x, const float* __restrict__ params,
                                                      const AugStat* __restrict__ st, float* __restrict__ part, size_t spatial) {
  __shared__ float scratch[4];
  const float* p = x + (size_t)blockIdx.y * spatial;
  const float* pr = params + (size_t)blockIdx.y * 3;
  const AugStat s = st[blockIdx.y];
  float sum = 0.f;
  const size_t v0 = (size_t)blockIdx.x * AUG_BLOCK;
  for (int it = 0; it < 8; ++it) {
    const size_t v = v0 + (size_t)it * 256 + threadIdx.x;
    if (v < spatial) sum += gamma_map(p[v], pr[0], s.gmin, s.grange, pr[1]);
  }
  sum = block_sum<4>(sum, scratch);
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = sum;
}

__global__ __launch_bounds__(64) void aug_finalize2_kernel(const float* __restrict__ part, AugStat* __restrict__ st, int nblocks,
                                                           double count) {
  double a = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) a += (double)part[(size_t)blockIdx.x * nblocks + i];
  a = wave_sum(a);
  if (threadIdx.x == 0) st[blockIdx.x].mean = (float)(a / count);
}

__global__ __launch_bounds__(256) void aug_apply_kernel(float* __restrict__ x, const float* __restrict__ params,
                                                        const AugStat* __restrict__ st, size_t spatial) {
  float* p = x + (size_t)blockIdx.y * spatial;
  const float* pr = params + (size_t)blockIdx.y * 3;
  const AugStat s = st[blockIdx.y];
  const float add = pr[0], gamma = pr[1], factor = pr[2];
  const size_t v0 = (size_t)blockIdx.x * AUG_BLOCK;
  for (int it = 0; it < 8; ++it) {
    const size_t v = v0 + (size_t)it * 256 + threadIdx.x;
    if (v < spatial) {
      const float g = gamma_map(p[v], add, s.gmin, s.grange, gamma);
      p[v] = fminf(fmaxf((g - s.mean) * factor + s.mean, s.lo), s.hi);
    }
  }
}

}  // namespace mednet

using namespace mednet;

static inline unsigned aug_blocks(size_t spatial) { return (unsigned)((spatial + AUG_BLOCK - 1) / AUG_BLOCK); }

extern "C" size_t mednet_augment_ws_bytes(int batch, int channels, size_t spatial) {
  return ((size_t)batch * channels * aug_blocks(spatial) * 2 + (size_t)batch * channels * 8 + 64) * sizeof(float);
}

extern "C" int mednet_augment_patches(float* data, const float* params, int batch, int channels, size_t spatial, void* ws,
                                      size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(batch > 0 && channels > 0 && channels <= 64 && spatial > 0, MEDNET_E_SHAPE,
                 "augment_patches: bad shape batch=%d channels=%d (1..64)", batch, channels);
  MEDNET_REQUIRE(ws_bytes >= mednet_augment_ws_bytes(batch, channels, spatial), MEDNET_E_WORKSPACE, "augment_patches: workspace too small");
  const unsigned nb = aug_blocks(spatial);
  float* part = (float*)ws;
  AugStat* st = (AugStat*)(part + (size_t)batch * channels * nb * 2);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(nb, batch * channels);
  hipLaunchKernelGGL(aug_minmax_kernel, grid, dim3(256), 0, s, data, part, spatial);
  hipLaunchKernelGGL(aug_finalize1_kernel, dim3(batch), dim3(64), 0, s, part, params, st, channels, (int)nb);
  hipLaunchKernelGGL(aug_sum_kernel, grid, dim3(256), 0, s, data, params, st, part, spatial);
  hipLaunchKernelGGL(aug_finalize2_kernel, dim3(batch * channels), dim3(64), 0, s, part, st, (int)nb, (double)spatial);
  hipLaunchKernelGGL(aug_apply_kernel, grid, dim3(256), 0, s, data, params, st, spatial);
  return check_launch("augment_patches");
}
