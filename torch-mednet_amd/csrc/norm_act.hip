// GroupNorm (+activation, +residual), stand-alone activations, 2x2x2 pooling, nearest-upsample+concat.
// All HBM-bound, channels-last: a lane reads 8 consecutive channels (16 B bf16 / 32 B fp32) of one voxel, a wave reads
// 64 such vectors back to back, i.e. fully coalesced 1-2 KiB per wave-instruction.  Reductions use wave64 shuffles,
// then LDS across the 4 waves of a workgroup, then a fixed-order fp64 combine of per-workgroup partials (no atomics:
// results are bitwise reproducible).
// Reference call sites: components.py:57 (GroupNorm), :36-40 (activations), :177-178 (residual add + activation),
// :208-212 (pooling), :277-280 (interpolate + cat).
#include "common.h"

#ifndef GN_NT
#define GN_NT 1  // GroupNorm backward's apply pass reads dz and y for the last time: non-temporal loads.  (Non-temporal STORES of
                 // the streaming kernels' outputs were measured too: +0.13 ms per step, profiles/r04_ab.md section 12.)
#endif
#ifndef GN_WAVES
#define GN_WAVES 3  // waves per SIMD the streaming GroupNorm kernels are compiled for (register cap 512 / GN_WAVES).  At 4
                    // the 8-wide backward kernels spilled 3-13 registers to scratch; A/B of the training step (same box, three
                    // interleaved rounds, profiles/r03_ab.md): 4 / 3 / 2 waves all 21.5-21.6 ms -- so the setting without scratch
#endif
namespace mednet {

// "column persistent" thread layout over a channels-last tensor: a thread keeps the same VEC channels for its
// whole life, so per-channel coefficients live in registers and per-channel sums need no atomics.
template <int VEC>
struct Cols {
  int cols, rows, col, row;
  bool active;
  __device__ __forceinline__ Cols(int c) {
    cols = c / VEC;
    rows = 256 / cols;
    active = (int)threadIdx.x < rows * cols;
    col = threadIdx.x % cols;
    row = threadIdx.x / cols;
  }
};

template <typename T, int VEC>
struct VecIO;
template <typename T>
struct VecIO<T, 8> {
  static __device__ __forceinline__ F8 load(const T* p, size_t i) { return ld8(p, i); }
  static __device__ __forceinline__ F8 load_last(const T* p, size_t i) { return GN_NT ? ld8_nt(p, i) : ld8(p, i); }
  static __device__ __forceinline__ void store(T* p, size_t i, const F8& v) { st8(p, i, v); }
};
// fp32 storage: 4 channels = one 16-byte access per lane (8 columns per 32-channel row: a wave instruction covers whole
// 128-byte rows); the 8-wide form needs two accesses per tensor and voxel and twice the registers (it spilled at 4 waves/SIMD)
template <>
struct VecIO<float, 4> {
  static __device__ __forceinline__ F8 load(const float* p, size_t i) {
    F8 r;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) r.v[k] = a[k];
    return r;
  }
  static __device__ __forceinline__ F8 load_last(const float* p, size_t i) {
    F8 r;
    const f32x4 a = GN_NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i)) : *reinterpret_cast<const f32x4*>(p + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) r.v[k] = a[k];
    return r;
  }
  static __device__ __forceinline__ void store(float* p, size_t i, const F8& v) {
    const f32x4 a = {v.v[0], v.v[1], v.v[2], v.v[3]};
    *reinterpret_cast<f32x4*>(p + i) = a;
  }
};
template <typename T>
struct VecIO<T, 1> {
  static __device__ __forceinline__ F8 load(const T* p, size_t i) {
    F8 r;
    r.v[0] = ld(p, i);
    return r;
  }
  static __device__ __forceinline__ F8 load_last(const T* p, size_t i) { return load(p, i); }
  static __device__ __forceinline__ void store(T* p, size_t i, const F8& v) { st(p, i, v.v[0]); }
};

static inline void chunk_plan(size_t spatial, int c, int vec, size_t& chunk_vox, unsigned& chunks) {
  const int cols = c / vec;
  const int rows = 256 / cols;
  size_t cv = (spatial + 1023) / 1024;
  if (cv < (size_t)rows * 8) cv = (size_t)rows * 8;
  cv = (cv + rows - 1) / rows * rows;
  chunk_vox = cv;
  chunks = (unsigned)((spatial + cv - 1) / cv);
}

// Sum, across the workgroup, the per-thread values of threads that share a column. vals[k] (k < NV*VEC) per thread.
// Result for (col, j) lands in out[col*NV*VEC + j] written by the threads of row 0.
template <int NVAL>
__device__ __forceinline__ void column_reduce(const float* vals, int cols, int rows, int col, int row, bool active,
                                              float* lds, float* out) {
  // lds: [256][NVAL]
  __syncthreads();
  if (active) {
#pragma unroll
    for (int k = 0; k < NVAL; ++k) lds[(row * cols + col) * NVAL + k] = vals[k];
  }
  __syncthreads();
  if (active && row == 0) {
#pragma unroll
    for (int k = 0; k < NVAL; ++k) {
      float s = 0.f;
      for (int r = 0; r < rows; ++r) s += lds[(r * cols + col) * NVAL + k];
      out[col * NVAL + k] = s;
    }
  }
}

// The same without LDS (see common.h, "reductions": a kernel that touches the LDS queue stalls next to the weight-gradient
// kernel on the side stream): possible when the column count is a power of two <= 64 (then the threads of a wave that share
// a column are the lanes with equal lane % cols: DPP / permlane sums) or a multiple of 64 (a wave's lanes are distinct
// columns).  Each WAVE writes its own partial row, so a workgroup yields rows_per_wg(cols) rows instead of one.
static inline int lds_free_rows_per_wg(int cols) {
  if (cols <= 64) return (cols & (cols - 1)) == 0 ? 4 : 0;  // 0: not applicable, use column_reduce
  return cols % 64 == 0 && 256 % cols == 0 ? 256 / cols : 0;
}
template <int NVAL>
__device__ __forceinline__ void column_reduce_lds_free(float* vals, int cols, int col, bool active, float* out_rows, int c2) {
  // out_rows: first of this workgroup's rows (row stride c2 = 2 * C floats); a thread's NVAL values are consecutive in a row
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (cols <= 64) {
#pragma unroll
    for (int k = 0; k < NVAL; ++k) {
      float v = active ? vals[k] : 0.f;
      switch (cols) {
        case 1: v = lane_class_sum<1>(v); break;
        case 2: v = lane_class_sum<2>(v); break;
        case 4: v = lane_class_sum<4>(v); break;
        case 8: v = lane_class_sum<8>(v); break;
        case 16: v = lane_class_sum<16>(v); break;
        case 32: v = lane_class_sum<32>(v); break;
        default: break;
      }
      vals[k] = v;
    }
    if (lane < cols) {
      float* o = out_rows + (size_t)wv * c2 + (size_t)lane * NVAL;
#pragma unroll
      for (int k = 0; k < NVAL; ++k) o[k] = vals[k];
    }
  } else if (active) {
    float* o = out_rows + (size_t)(threadIdx.x / cols) * c2 + (size_t)col * NVAL;
#pragma unroll
    for (int k = 0; k < NVAL; ++k) o[k] = vals[k];
  }
}

// ---------------------------------------------------------------------------------------------- GN statistics
// partial[n][chunk][c][2] = {sum x, sum x^2}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                         size_t spatial, int c, size_t chunk_vox) {
  __shared__ float lds[256 * 2 * VEC];
  const Cols<VEC> L(c);
  const int n = blockIdx.y;
  const size_t v0 = (size_t)blockIdx.x * chunk_vox;
  const size_t v1 = v0 + chunk_vox < spatial ? v0 + chunk_vox : spatial;
  float acc[2 * VEC];
#pragma unroll
  for (int k = 0; k < 2 * VEC; ++k) acc[k] = 0.f;
  if (L.active) {
    const T* base = x + (size_t)n * spatial * c + (size_t)L.col * VEC;
    size_t v = v0 + L.row;
    for (; v + 3 * (size_t)L.rows < v1; v += 4 * (size_t)L.rows) {  // 4 independent 16-byte loads in flight per lane
      F8 xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[u] = VecIO<T, VEC>::load(base, (v + (size_t)u * L.rows) * c);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          acc[2 * k] += xv[u].v[k];
          acc[2 * k + 1] = fmaf(xv[u].v[k], xv[u].v[k], acc[2 * k + 1]);
        }
    }
    for (; v < v1; v += L.rows) {
      const F8 xv = VecIO<T, VEC>::load(base, v * c);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        acc[2 * k] += xv.v[k];
        acc[2 * k + 1] = fmaf(xv.v[k], xv.v[k], acc[2 * k + 1]);
      }
    }
  }
  float* out = partial + ((size_t)n * gridDim.x + blockIdx.x) * c * 2;
  column_reduce<2 * VEC>(acc, L.cols, L.rows, L.col, L.row, L.active, lds, out);
}

// stage 1 of every finalize: csum[n][c][2] = sum over chunks of partial[n][chunk][c][2]; one wave per (n, c), lanes stride
// the chunks (fp64, fixed order).
__global__ __launch_bounds__(64) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ csum,
                                                             int c, int chunks) {
  const int n = blockIdx.x / c, cc = blockIdx.x % c;
  double a = 0.0, b = 0.0;
  for (int ch = threadIdx.x; ch < chunks; ch += 64) {
    const float* p = partial + (((size_t)n * chunks + ch) * c + cc) * 2;
    a += (double)p[0];
    b += (double)p[1];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if (threadIdx.x == 0) {
    csum[((size_t)n * c + cc) * 2] = (float)a;
    csum[((size_t)n * c + cc) * 2 + 1] = (float)b;
  }
}

// the same for per-channel {sum du, sum du * x} rows written by the data-gradient conv's epilogue (conv_mfma GNB):
// csum[n][c] = {sum du, sum du * xhat} with sum du * xhat = rstd * (sum du * x - mean * sum du)
__global__ __launch_bounds__(64) void reduce_partials_dux_kernel(const float* __restrict__ partial,
                                                                 const float* __restrict__ stats, float* __restrict__ csum,
                                                                 int c, int groups, int chunks) {
  const int n = blockIdx.x / c, cc = blockIdx.x % c;
  double a = 0.0, b = 0.0, a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;  // four row loads in flight
  const float* base = partial + ((size_t)n * chunks * c + cc) * 2;
  const size_t rs = (size_t)c * 2;
  int ch = threadIdx.x;
  for (; ch + 192 < chunks; ch += 256) {
    const float2 p0 = *reinterpret_cast<const float2*>(base + ch * rs), p1 = *reinterpret_cast<const float2*>(base + (ch + 64) * rs);
    const float2 p2 = *reinterpret_cast<const float2*>(base + (ch + 128) * rs), p3 = *reinterpret_cast<const float2*>(base + (ch + 192) * rs);
    a += (double)p0.x;
    b += (double)p0.y;
    a1 += (double)p1.x;
    b1 += (double)p1.y;
    a2 += (double)p2.x;
    b2 += (double)p2.y;
    a3 += (double)p3.x;
    b3 += (double)p3.y;
  }
  for (; ch < chunks; ch += 64) {
    const float2 p0 = *reinterpret_cast<const float2*>(base + ch * rs);
    a += (double)p0.x;
    b += (double)p0.y;
  }
  a = wave_sum((a + a1) + (a2 + a3));
  b = wave_sum((b + b1) + (b2 + b3));
  if (threadIdx.x == 0) {
    const int g = cc / (c / groups);
    const double mean = stats[((size_t)n * groups + g) * 2], rstd = stats[((size_t)n * groups + g) * 2 + 1];
    csum[((size_t)n * c + cc) * 2] = (float)a;
    csum[((size_t)n * c + cc) * 2 + 1] = (float)(rstd * (b - mean * a));
  }
}

// one workgroup per (n, g) -- 256 threads, or 1024 when there are thousands of rows (the first layer writes 16 384 per sample: a
// thread's chain of dependent round trips, four rows in flight, was 46 us of exposed latency) --: a thread walks rows t, t + NT,
// ... and adds the group's cg {sum, sumsq} pairs of each row (32 contiguous bytes when cg = 4); fp64, fixed order => bitwise
// reproducible; then mean, rstd and the per-channel affine the apply kernel uses.
__global__ __launch_bounds__(1024) void gn_finalize_kernel(const float* __restrict__ partial,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ stats,
                                                           float* __restrict__ coef, int c, int groups, int chunks,
                                                           double count, float eps) {
  __shared__ double sh[2][16];
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cg = c / groups, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nt = (int)blockDim.x, nwv = nt >> 6;
  double s = 0.0, q = 0.0;
  const float* base = partial + (size_t)n * chunks * c * 2 + (size_t)g * cg * 2;
  const size_t row = (size_t)c * 2;
  if (cg % 2 == 0) {
    auto add_row = [&](const float* p, double& ss, double& qq) {
      for (int i = 0; i < cg / 2; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i);
        ss += (double)v[0] + (double)v[2];
        qq += (double)v[1] + (double)v[3];
      }
    };
    double s1 = 0.0, q1 = 0.0, s2 = 0.0, q2 = 0.0, s3 = 0.0, q3 = 0.0;
    int ch = threadIdx.x;
    for (; ch + 3 * nt < chunks; ch += 4 * nt) {
      add_row(base + (size_t)ch * row, s, q);
      add_row(base + (size_t)(ch + nt) * row, s1, q1);
      add_row(base + (size_t)(ch + 2 * nt) * row, s2, q2);
      add_row(base + (size_t)(ch + 3 * nt) * row, s3, q3);
    }
    for (; ch < chunks; ch += nt) add_row(base + (size_t)ch * row, s, q);
    s = (s + s1) + (s2 + s3);
    q = (q + q1) + (q2 + q3);
  } else {
    for (int ch = threadIdx.x; ch < chunks; ch += nt) {
      const float* p = base + (size_t)ch * row;
      for (int i = 0; i < cg; ++i) {
        s += (double)p[2 * i];
        q += (double)p[2 * i + 1];
      }
    }
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if (lane == 0) {
    sh[0][wv] = s;
    sh[1][wv] = q;
  }
  __syncthreads();
  s = q = 0.0;
  for (int k = 0; k < nwv; ++k) {  // (fixed order)
    s += sh[0][k];
    q += sh[1][k];
  }
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    stats[((size_t)n * groups + g) * 2] = (float)mean;
    stats[((size_t)n * groups + g) * 2 + 1] = rstd;
  }
  for (int i = threadIdx.x; i < cg; i += nt) {
    const int cc = g * cg + i;
    const float ga = gamma ? gamma[cc] : 1.f, be = beta ? beta[cc] : 0.f;
    const float a = ga * rstd;
    coef[((size_t)n * c + cc) * 2] = a;
    coef[((size_t)n * c + cc) * 2 + 1] = be - (float)mean * a;
  }
}

// ---------------------------------------------------------------------------------------------- GN apply (+act, +residual)
template <typename TX, typename TZ, int VEC>
__global__ __launch_bounds__(256) void gn_act_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ coef,
                                                         const TZ* __restrict__ res, TZ* __restrict__ z,
                                                         size_t spatial, int c, int act, size_t chunk_vox) {
  const Cols<VEC> L(c);
  if (!L.active) return;
  const int n = blockIdx.y;
  const size_t v0 = (size_t)blockIdx.x * chunk_vox;
  const size_t v1 = v0 + chunk_vox < spatial ? v0 + chunk_vox : spatial;
  float a[VEC], b[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    a[k] = coef[((size_t)n * c + L.col * VEC + k) * 2];
    b[k] = coef[((size_t)n * c + L.col * VEC + k) * 2 + 1];
  }
  const size_t off = (size_t)n * spatial * c + (size_t)L.col * VEC;
  auto one = [&](const F8& xin, const F8& rin, F8& out) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      out.v[k] = fmaf(a[k], xin.v[k], b[k]);
      if (res) out.v[k] += rin.v[k];
    }
    act_apply_n<VEC>(out.v, act);
  };
  size_t v = v0 + L.row;
  const size_t R = (size_t)L.rows;
  for (; v + 3 * R < v1; v += 4 * R) {  // four voxels per trip: 4-8 loads in flight per lane
    size_t i[4];
    F8 xi[4], ri[4], zo[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) i[u] = off + (v + u * R) * c;
#pragma unroll
    for (int u = 0; u < 4; ++u) xi[u] = VecIO<TX, VEC>::load(x, i[u]);
    if (res) {
#pragma unroll
      for (int u = 0; u < 4; ++u) ri[u] = VecIO<TZ, VEC>::load(res, i[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(xi[u], ri[u], zo[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) VecIO<TZ, VEC>::store(z, i[u], zo[u]);
  }
  for (; v < v1; v += R) {
    const size_t i = off + v * c;
    const F8 xv = VecIO<TX, VEC>::load(x, i);
    F8 rv, zv;
    if (res) rv = VecIO<TZ, VEC>::load(res, i);
    one(xv, rv, zv);
    VecIO<TZ, VEC>::store(z, i, zv);
  }
}

// ---------------------------------------------------------------------------------------------- GN backward
// pass 1: partial[n][chunk][c][2] = {sum du, sum du * xhat},  du = (dz + dz2) * act'(z)
template <typename T, int VEC>
__global__ __launch_bounds__(256, GN_WAVES) void gn_bwd_partial_kernel(const T* __restrict__ dz, const T* __restrict__ dz2,
                                                             const T* __restrict__ x, const T* __restrict__ z,
                                                             const float* __restrict__ coef,
                                                             const float* __restrict__ stats,
                                                             float* __restrict__ partial, size_t spatial, int c,
                                                             int groups, int act, size_t chunk_vox, int rpw) {
  __shared__ float lds[256 * 2 * VEC];
  const Cols<VEC> L(c);
  const int n = blockIdx.y;
  const size_t v0 = (size_t)blockIdx.x * chunk_vox;
  const size_t v1 = v0 + chunk_vox < spatial ? v0 + chunk_vox : spatial;
  float acc[2 * VEC];
#pragma unroll
  for (int k = 0; k < 2 * VEC; ++k) acc[k] = 0.f;
  if (L.active) {
    const int cg = c / groups;
    float mean[VEC], rstd[VEC], ca[VEC], cb[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int g = (L.col * VEC + k) / cg;
      mean[k] = stats[((size_t)n * groups + g) * 2];
      rstd[k] = stats[((size_t)n * groups + g) * 2 + 1];
      ca[k] = z ? 0.f : coef[((size_t)n * c + L.col * VEC + k) * 2];
      cb[k] = z ? 0.f : coef[((size_t)n * c + L.col * VEC + k) * 2 + 1];
    }
    const size_t off = (size_t)n * spatial * c + (size_t)L.col * VEC;
    auto one = [&](size_t i) {
      F8 g1 = VecIO<T, VEC>::load(dz, i);
      F8 g2, zv;
      if (dz2) g2 = VecIO<T, VEC>::load(dz2, i);
      if (act != MEDNET_ACT_NONE && z) zv = VecIO<T, VEC>::load(z, i);
      const F8 xv = VecIO<T, VEC>::load(x, i);
      if (dz2) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) g1.v[k] += g2.v[k];
      }
      if (z) {
        act_grad_n<VEC>(g1.v, zv.v, act);
      } else {  // the activated tensor is not read back: its pre-activation is one FMA away from the conv output
#pragma unroll
        for (int k = 0; k < VEC; ++k) zv.v[k] = fmaf(ca[k], xv.v[k], cb[k]);
        act_grad_pre_n<VEC>(g1.v, zv.v, act);
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float xh = (xv.v[k] - mean[k]) * rstd[k];
        acc[2 * k] += g1.v[k];
        acc[2 * k + 1] = fmaf(g1.v[k], xh, acc[2 * k + 1]);
      }
    };
    size_t v = v0 + L.row;
    for (; v + (size_t)L.rows < v1; v += 2 * (size_t)L.rows) {  // (four per trip measured 7 % slower)
      one(off + v * c);
      one(off + (v + L.rows) * c);
    }
    for (; v < v1; v += L.rows) one(off + v * c);
  }
  if (rpw > 0) {  // rpw rows per workgroup, no LDS
    float* out = partial + ((size_t)n * gridDim.x + blockIdx.x) * rpw * c * 2;
    column_reduce_lds_free<2 * VEC>(acc, L.cols, L.col, L.active, out, c * 2);
    return;
  }
  float* out = partial + ((size_t)n * gridDim.x + blockIdx.x) * c * 2;
  column_reduce<2 * VEC>(acc, L.cols, L.rows, L.col, L.row, L.active, lds, out);
}

// pass 2a: per (n,g): S1 = sum_c gamma_c * sum du_c, S2 = sum_c gamma_c * sum(du*xhat)_c  ->
//   dx = k1_c * du + k2_g * x + k3_g,  k1 = rstd*gamma_c, k2 = -rstd^2*S2/M, k3 = (rstd^2*S2*mean - rstd*S1)/M
// csum[n][c][2] (from reduce_partials_kernel) holds the per-(n,c) totals.
__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(const float* __restrict__ csum,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ gamma,
                                                             float* __restrict__ bcoef, int c, int groups, double count) {
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cg = c / groups;
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < cg; i += 64) {
    const int cc = g * cg + i;
    const double ga = gamma ? (double)gamma[cc] : 1.0;
    s1 += ga * (double)csum[((size_t)n * c + cc) * 2];
    s2 += ga * (double)csum[((size_t)n * c + cc) * 2 + 1];
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const double mean = stats[((size_t)n * groups + g) * 2], rstd = stats[((size_t)n * groups + g) * 2 + 1];
  const float k2 = (float)(-rstd * rstd * s2 / count);
  const float k3 = (float)((rstd * rstd * s2 * mean - rstd * s1) / count);
  for (int i = threadIdx.x; i < cg; i += 64) {
    const int cc = g * cg + i;
    float* o = bcoef + ((size_t)n * c + cc) * 3;
    o[0] = (float)rstd * (gamma ? gamma[cc] : 1.f);
    o[1] = k2;
    o[2] = k3;
  }
}
// passes 1b + 2a in ONE launch when a group's channel count divides 256 (every GroupNorm of the U-Nets here): a workgroup per
// (n, g) sums the group's columns of the partial rows -- thread t always works on channel t % cg, rows t / cg, t / cg + 256 / cg,
// ... four loads in flight, fp64, fixed order -- and goes on to S1, S2 and the coefficients.  As three launches (reduce, finalize,
// parameter gradients: 18 us + three launch gaps per GroupNorm, 21 GroupNorms per step) this was 0.4 ms of the 22 ms step; the
// parameter gradients ride in the apply kernel's first workgroup (GnParamGrad).
// DUX: the rows hold {sum du, sum du * x} (conv epilogues) instead of {sum du, sum du * xhat} (gn_bwd_partial_kernel).
template <bool DUX>
__global__ __launch_bounds__(256) void gn_bwd_reduce_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ stats,
                                                                     const float* __restrict__ gamma, float* __restrict__ csum,
                                                                     float* __restrict__ bcoef, int c, int groups, int rows,
                                                                     double count) {
  __shared__ double sha[256], shb[256];
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cg = c / groups, t = threadIdx.x;
  const int i = t % cg, r0 = t / cg, rstep = 256 / cg;
  const float* base = partial + (size_t)n * rows * c * 2 + (size_t)(g * cg + i) * 2;
  const size_t rs = (size_t)c * 2;
  // (16 rows in flight per thread: with 4, the 2 048 rows a ConvTranspose data gradient writes per sample were 16 dependent round
  //  trips, 45 us of a kernel that otherwise takes 7)
  constexpr int U = 16;
  double a[U], b[U];
#pragma unroll
  for (int k = 0; k < U; ++k) a[k] = b[k] = 0.0;
  int r = r0;
  for (; r + (U - 1) * rstep < rows; r += U * rstep) {
    float2 p[U];
#pragma unroll
    for (int k = 0; k < U; ++k) p[k] = *reinterpret_cast<const float2*>(base + (size_t)(r + k * rstep) * rs);
#pragma unroll
    for (int k = 0; k < U; ++k) {
      a[k] += (double)p[k].x;
      b[k] += (double)p[k].y;
    }
  }
  for (; r < rows; r += rstep) {
    const float2 p0 = *reinterpret_cast<const float2*>(base + (size_t)r * rs);
    a[0] += (double)p0.x;
    b[0] += (double)p0.y;
  }
#pragma unroll
  for (int w = 1; w < U; w *= 2) {  // fixed pairwise order
#pragma unroll
    for (int k = 0; k < U; k += 2 * w) {
      a[k] += a[k + w];
      b[k] += b[k + w];
    }
  }
  sha[t] = a[0];
  shb[t] = b[0];
  __syncthreads();
  const double mean = stats[((size_t)n * groups + g) * 2], rstd = stats[((size_t)n * groups + g) * 2 + 1];
  if (t < cg) {  // channel t of the group: its 256 / cg row classes, in order
    double a = 0.0, b = 0.0;
    for (int k = 0; k < rstep; ++k) {
      a += sha[k * cg + t];
      b += shb[k * cg + t];
    }
    if (DUX) b = rstd * (b - mean * a);
    const float af = (float)a, bf = (float)b;  // (the finalize below reads what csum holds, as the three-launch form did)
    csum[((size_t)n * c + g * cg + t) * 2] = af;
    csum[((size_t)n * c + g * cg + t) * 2 + 1] = bf;
    const double ga = gamma ? (double)gamma[g * cg + t] : 1.0;
    sha[t] = ga * (double)af;
    shb[t] = ga * (double)bf;
  }
  __syncthreads();
  if (t < cg) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < cg; ++k) {
      s1 += sha[k];
      s2 += shb[k];
    }
    float* o = bcoef + ((size_t)n * c + g * cg + t) * 3;
    o[0] = (float)rstd * (gamma ? gamma[g * cg + t] : 1.f);
    o[1] = (float)(-rstd * rstd * s2 / count);
    o[2] = (float)((rstd * rstd * s2 * mean - rstd * s1) / count);
  }
}
static bool gn_bwd_one_launch(int c, int groups) {
  const int cg = c / groups;
  return cg >= 1 && cg <= 256 && 256 % cg == 0 && tuning_option("gn_bwd_one_launch", 1);
}
// the parameter gradients of pass 2b, computed by the first workgroup of the apply kernel when the one-launch form ran
struct GnParamGrad {
  const float* csum = nullptr;
  float* dgamma = nullptr;
  float* dbeta = nullptr;
  int n = 0;
};
// pass 2b: dgamma_c = sum_n sum(du*xhat), dbeta_c = sum_n sum du
__global__ __launch_bounds__(256) void gn_bwd_params_kernel(const float* __restrict__ csum, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int n, int c) {
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc >= c) return;
  double a = 0.0, b = 0.0;
  for (int i = 0; i < n; ++i) {
    a += (double)csum[((size_t)i * c + cc) * 2];
    b += (double)csum[((size_t)i * c + cc) * 2 + 1];
  }
  if (dbeta) dbeta[cc] = (float)a;
  if (dgamma) dgamma[cc] = (float)b;
}
// pass 3: apply
template <typename T, int VEC>
__global__ __launch_bounds__(256, GN_WAVES) void gn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ dz2,
                                                           const T* __restrict__ x, const T* __restrict__ z,
                                                           const float* __restrict__ coef,
                                                           const float* __restrict__ bcoef, T* __restrict__ dx,
                                                           T* __restrict__ dres, size_t spatial, int c, int act,
                                                           size_t chunk_vox, int in_act, GnParamGrad pg) {
  if (pg.csum && blockIdx.x == 0 && blockIdx.y == 0) {  // pass 2b (gn_bwd_params_kernel) without a launch of its own
    for (int cc = threadIdx.x; cc < c; cc += 256) {
      double a = 0.0, b = 0.0;
      for (int i = 0; i < pg.n; ++i) {
        a += (double)pg.csum[((size_t)i * c + cc) * 2];
        b += (double)pg.csum[((size_t)i * c + cc) * 2 + 1];
      }
      if (pg.dbeta) pg.dbeta[cc] = (float)a;
      if (pg.dgamma) pg.dgamma[cc] = (float)b;
    }
  }
  const Cols<VEC> L(c);
  if (!L.active) return;
  const int n = blockIdx.y;
  const size_t v0 = (size_t)blockIdx.x * chunk_vox;
  const size_t v1 = v0 + chunk_vox < spatial ? v0 + chunk_vox : spatial;
  float k1[VEC], k2[VEC], k3[VEC], ca[VEC], cb[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const float* o = bcoef + ((size_t)n * c + L.col * VEC + k) * 3;
    k1[k] = o[0];
    k2[k] = o[1];
    k3[k] = o[2];
    ca[k] = z ? 0.f : coef[((size_t)n * c + L.col * VEC + k) * 2];
    cb[k] = z ? 0.f : coef[((size_t)n * c + L.col * VEC + k) * 2 + 1];
  }
  const size_t off = (size_t)n * spatial * c + (size_t)L.col * VEC;
  auto one = [&](size_t i) {
    F8 g1 = VecIO<T, VEC>::load_last(dz, i);
    F8 g2, zv;
    if (dz2) g2 = VecIO<T, VEC>::load_last(dz2, i);
    if (act != MEDNET_ACT_NONE && z) zv = VecIO<T, VEC>::load(z, i);
    const F8 xv = VecIO<T, VEC>::load_last(x, i);
    if (dz2) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) g1.v[k] += g2.v[k];
    }
    if (z) {
      act_grad_n<VEC>(g1.v, zv.v, act);
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) zv.v[k] = fmaf(ca[k], xv.v[k], cb[k]);
      act_grad_pre_n<VEC>(g1.v, zv.v, act);
    }
    F8 o;
#pragma unroll
    for (int k = 0; k < VEC; ++k) o.v[k] = fmaf(k1[k], g1.v[k], fmaf(k2[k], xv.v[k], k3[k]));
    // x is the OUTPUT of an activation (conv -> ReLU -> this GroupNorm in the 'gcr' orders): its derivative is folded in
    // here, where x is in registers anyway, instead of a separate pass over (dx, x) in the conv layer's backward
    if (in_act != MEDNET_ACT_NONE) act_grad_n<VEC>(o.v, xv.v, in_act);
    VecIO<T, VEC>::store(dx, i, o);
    if (dres) VecIO<T, VEC>::store(dres, i, g1);
  };
  size_t v = v0 + L.row;
  for (; v + (size_t)L.rows < v1; v += 2 * (size_t)L.rows) {
    one(off + v * c);
    one(off + (v + L.rows) * c);
  }
  for (; v < v1; v += L.rows) one(off + v * c);
}

// ---------------------------------------------------------------------------------------------- flat elementwise
template <typename T>
__global__ __launch_bounds__(256) void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ z, size_t count, int act) {
  const size_t stride = (size_t)gridDim.x * 256 * 8;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8; i < count; i += stride) {
    if (i + 8 <= count) {
      F8 v = ld8(x, i);
#pragma unroll
      for (int k = 0; k < 8; ++k) v.v[k] = act_apply(v.v[k], act);
      st8(z, i, v);
    } else {
      for (size_t j = i; j < count; ++j) st(z, j, act_apply(ld(x, j), act));
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ z,
                                                      T* __restrict__ dx, size_t count, int act) {
  const size_t stride = (size_t)gridDim.x * 256 * 8;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8; i < count; i += stride) {
    if (i + 8 <= count) {
      F8 g = ld8(dz, i);
      const F8 zv = ld8(z, i);
#pragma unroll
      for (int k = 0; k < 8; ++k) g.v[k] *= act_grad_from_out(zv.v[k], act);
      st8(dx, i, g);
    } else {
      for (size_t j = i; j < count; ++j) st(dx, j, ld(dz, j) * act_grad_from_out(ld(z, j), act));
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o,
                                                  size_t count) {
  const size_t stride = (size_t)gridDim.x * 256 * 8;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8; i < count; i += stride) {
    if (i + 8 <= count) {
      F8 u = ld8(a, i);
      const F8 w = ld8(b, i);
#pragma unroll
      for (int k = 0; k < 8; ++k) u.v[k] += w.v[k];
      st8(o, i, u);
    } else {
      for (size_t j = i; j < count; ++j) st(o, j, ld(a, j) + ld(b, j));
    }
  }
}

// ---------------------------------------------------------------------------------------------- 2x2x2 pooling
template <typename T, int VEC>
__global__ __launch_bounds__(256) void pool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int d, int h,
                                                        int w, int c, int mode) {
  const int od = d / 2, oh = h / 2, ow = w / 2, cv = c / VEC;
  const size_t total = (size_t)n * od * oh * ow * cv;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cc = (int)(i % cv);
  size_t v = i / cv;
  const int ox = (int)(v % ow);
  v /= ow;
  const int oy = (int)(v % oh);
  v /= oh;
  const int oz = (int)(v % od);
  const int nn = (int)(v / od);
  F8 best;
#pragma unroll
  for (int k = 0; k < VEC; ++k) best.v[k] = mode == MEDNET_POOL_MAX ? -INFINITY : 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int iz = 2 * oz + (t >> 2), iy = 2 * oy + ((t >> 1) & 1), ix = 2 * ox + (t & 1);
    const size_t src = ((((size_t)nn * d + iz) * h + iy) * w + ix) * c + (size_t)cc * VEC;
    const F8 xv = VecIO<T, VEC>::load(x, src);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      if (mode == MEDNET_POOL_MAX) best.v[k] = (xv.v[k] > best.v[k] || xv.v[k] != xv.v[k]) ? xv.v[k] : best.v[k];
      else best.v[k] += xv.v[k];
    }
  }
  if (mode == MEDNET_POOL_AVG) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) best.v[k] *= 0.125f;
  }
  VecIO<T, VEC>::store(y, i * VEC, best);
}

// GroupNorm apply (+ residual, + activation) of an encoder block's last layer AND the 2x2x2 pooling that consumes the block's
// output (components.py:177-178 -> :222-224), in one pass: a thread owns one POOLED voxel and 8 channels, normalises the 8 voxels of
// its window (reads y and the residual, writes the block output z) and writes their maximum / mean to the pooled tensor -- the
// stand-alone pooling pass and its re-read of z are gone.  z is computed and rounded exactly as gn_act_fwd_kernel
// does, the pooling runs on the ROUNDED values in pool2_fwd_kernel's scan order: both outputs are bit-identical to the two
// launches.  Even d, h, w.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_act_pool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ coef,
                                                              const T* __restrict__ res, T* __restrict__ z, T* __restrict__ pooled,
                                                              int n, int d, int h, int w, int c, int act, int mode) {
  const int od = d / 2, oh = h / 2, ow = w / 2, cv = c / VEC;
  const size_t total = (size_t)n * od * oh * ow * cv;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cc = (int)(i % cv);
  size_t v = i / cv;
  const int ox = (int)(v % ow);
  v /= ow;
  const int oy = (int)(v % oh);
  v /= oh;
  const int oz = (int)(v % od);
  const int nn = (int)(v / od);
  float a[VEC], b[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    a[k] = coef[((size_t)nn * c + cc * VEC + k) * 2];
    b[k] = coef[((size_t)nn * c + cc * VEC + k) * 2 + 1];
  }
  size_t src[8];
  F8 xv[8], rv[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int iz = 2 * oz + (t >> 2), iy = 2 * oy + ((t >> 1) & 1), ix = 2 * ox + (t & 1);
    src[t] = ((((size_t)nn * d + iz) * h + iy) * w + ix) * c + (size_t)cc * VEC;
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) xv[t] = VecIO<T, VEC>::load(x, src[t]);
  if (res) {
#pragma unroll
    for (int t = 0; t < 8; ++t) rv[t] = VecIO<T, VEC>::load(res, src[t]);
  }
  F8 best;
#pragma unroll
  for (int k = 0; k < VEC; ++k) best.v[k] = mode == MEDNET_POOL_MAX ? -INFINITY : 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    F8 o;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      o.v[k] = fmaf(a[k], xv[t].v[k], b[k]);
      if (res) o.v[k] += rv[t].v[k];
    }
    act_apply_n<VEC>(o.v, act);
    VecIO<T, VEC>::store(z, src[t], o);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float s = (float)(T)o.v[k];  // what the pooling pass would read back
      if (mode == MEDNET_POOL_MAX) best.v[k] = (s > best.v[k] || s != s) ? s : best.v[k];
      else best.v[k] += s;
    }
  }
  if (mode == MEDNET_POOL_AVG) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) best.v[k] *= 0.125f;
  }
  VecIO<T, VEC>::store(pooled, i * VEC, best);
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void pool2_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                        const T* __restrict__ add, T* __restrict__ dx, int n, int d,
                                                        int h, int w, int c, int mode, int in_act, int add_c) {
  const int od = d / 2, oh = h / 2, ow = w / 2, cv = c / VEC;
  const size_t total = (size_t)n * od * oh * ow * cv;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cc = (int)(i % cv);
  size_t v = i / cv;
  const int ox = (int)(v % ow);
  v /= ow;
  const int oy = (int)(v % oh);
  v /= oh;
  const int oz = (int)(v % od);
  const int nn = (int)(v / od);
  const F8 g = VecIO<T, VEC>::load(dy, i * VEC);
  int arg[VEC];
  if (mode == MEDNET_POOL_MAX) {
    float best[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      best[k] = -INFINITY;
      arg[k] = 0;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int iz = 2 * oz + (t >> 2), iy = 2 * oy + ((t >> 1) & 1), ix = 2 * ox + (t & 1);
      const size_t src = ((((size_t)nn * d + iz) * h + iy) * w + ix) * c + (size_t)cc * VEC;
      const F8 xv = VecIO<T, VEC>::load(x, src);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if (xv.v[k] > best[k] || xv.v[k] != xv.v[k]) {  // first maximum in scan order, NaN propagates (ATen)
          best[k] = xv.v[k];
          arg[k] = t;
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int iz = 2 * oz + (t >> 2), iy = 2 * oy + ((t >> 1) & 1), ix = 2 * ox + (t & 1);
    const size_t dst = ((((size_t)nn * d + iz) * h + iy) * w + ix) * c + (size_t)cc * VEC;
    F8 o;
#pragma unroll
    for (int k = 0; k < VEC; ++k)
      o.v[k] = mode == MEDNET_POOL_MAX ? (arg[k] == t ? g.v[k] : 0.f) : 0.125f * g.v[k];
    if (add) {  // the other consumer's gradient of the same tensor (the decoder's skip join): one pass instead of an add kernel
      // (add_c: channels per voxel of the tensor `add` is a channel slice of -- the gradient of UNet3D's concatenation, read where
      //  it lies instead of being copied out first)
      const F8 a = VecIO<T, VEC>::load(add, (dst / c) * add_c + dst % c);
#pragma unroll
      for (int k = 0; k < VEC; ++k) o.v[k] += a.v[k];
    }
    // x is the OUTPUT of a fused conv -> activation layer (UNet3D's 'gcr' blocks, components.py:57-63): its derivative is folded in
    // here, on the value a stand-alone join would have stored, and the conv layer's backward skips its activation pass
    // (ops.ActMaskHook) -- 3 passes over the level's output less
    if (in_act != MEDNET_ACT_NONE) {
      const F8 xv = VecIO<T, VEC>::load(x, dst);
#pragma unroll
      for (int k = 0; k < VEC; ++k) o.v[k] = (float)(T)o.v[k];
      act_grad_n<VEC>(o.v, xv.v, in_act);
    }
    VecIO<T, VEC>::store(dx, dst, o);
  }
}

// The same join when x is the output of an ExtResNetBlock (an encoder level feeds the next level's pooling and the
// decoder's skip join, model.py:194-205): dx is then the gradient that block's backward starts from, and the first pass
// of its GroupNorm-3 backward -- du = dx * act'(x), per-channel sums of du and du * y3 (components.py:170-178) -- is taken
// here from the STORED dx, with x already in registers for the arg-max.  A thread walks PV pooled voxels; LDS-free column
// reduction, 4 partial rows per workgroup: partial[n][4 * workgroup + wave][c][2].  Even d, h, w only.
constexpr int POOL_GN_PV = 4;
template <typename T>
__global__ __launch_bounds__(256) void pool2_bwd_gn_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ add, T* __restrict__ dx,
                                                           const T* __restrict__ gy, float* __restrict__ partial, int d,
                                                           int h, int w, int c, int mode, int act) {
  constexpr int VEC = 8;
  const int od = d / 2, oh = h / 2, ow = w / 2, cv = c / VEC;
  const size_t per = (size_t)od * oh * ow * cv;
  const int nn = blockIdx.y;
  const int cc = (int)(threadIdx.x % cv);  // (256 % cv == 0: a thread keeps its channels)
  float acc[2 * VEC];
#pragma unroll
  for (int k = 0; k < 2 * VEC; ++k) acc[k] = 0.f;
  for (int it = 0; it < POOL_GN_PV; ++it) {
    const size_t i = ((size_t)blockIdx.x * POOL_GN_PV + it) * 256 + threadIdx.x;
    if (i >= per) break;
    size_t v = i / cv;
    const int ox = (int)(v % ow);
    v /= ow;
    const int oy = (int)(v % oh);
    const int oz = (int)(v / oh);
    const F8 g = VecIO<T, VEC>::load(dy, ((size_t)nn * per + i) * VEC);
    F8 xv[8];
    int arg[VEC];
    float best[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      best[k] = -INFINITY;
      arg[k] = 0;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int iz = 2 * oz + (t >> 2), iy = 2 * oy + ((t >> 1) & 1), ix = 2 * ox + (t & 1);
      const size_t src = ((((size_t)nn * d + iz) * h + iy) * w + ix) * c + (size_t)cc * VEC;
      xv[t] = VecIO<T, VEC>::load(x, src);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if (xv[t].v[k] > best[k] || xv[t].v[k] != xv[t].v[k]) {  // first maximum in scan order, NaN propagates (ATen)
          best[k] = xv[t].v[k];
          arg[k] = t;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int iz = 2 * oz + (t >> 2), iy = 2 * oy + ((t >> 1) & 1), ix = 2 * ox + (t & 1);
      const size_t dst = ((((size_t)nn * d + iz) * h + iy) * w + ix) * c + (size_t)cc * VEC;
      F8 o;
#pragma unroll
      for (int k = 0; k < VEC; ++k) o.v[k] = mode == MEDNET_POOL_MAX ? (arg[k] == t ? g.v[k] : 0.f) : 0.125f * g.v[k];
      if (add) {
        const F8 a = VecIO<T, VEC>::load(add, dst);
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] += a.v[k];
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) o.v[k] = (float)(T)o.v[k];  // the stored value is what GroupNorm-3's second pass reads
      if (dx) VecIO<T, VEC>::store(dx, dst, o);  // (dx == nullptr: statistics only, gn_bwd_apply_pool_kernel rebuilds the rows)
      const F8 yv = VecIO<T, VEC>::load(gy, dst);
      act_grad_n<VEC>(o.v, xv[t].v, act);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        acc[2 * k] += o.v[k];
        acc[2 * k + 1] = fmaf(o.v[k], yv.v[k], acc[2 * k + 1]);
      }
    }
  }
  float* out = partial + (((size_t)nn * gridDim.x + blockIdx.x) * 4) * c * 2;
  column_reduce_lds_free<2 * VEC>(acc, cv, cc, true, out, c * 2);
}

// GroupNorm-3's backward apply pass of an encoder block whose output gradient is the pooling backward + skip join above, WITHOUT
// that gradient in memory: a thread owns one pooled voxel x VEC channels, rebuilds the 8 gradient rows of its window from the
// pooled gradient, the arg-max of the block output (which it reads anyway for act') and the skip gradient -- rounded to T exactly
// as pool2_bwd_gn_kernel rounds what it would have stored -- and applies gn_bwd_apply_kernel's closed form:
//   du = dz * act'(out);  dy3 = k1 * du + k2 * y3 + k3;  dres = du.
// The 537 MB gradient tensor of a 32-channel 128^3 batch is neither written (pooling backward) nor read (this pass).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_bwd_apply_pool_kernel(const T* __restrict__ dyp, const T* __restrict__ add,
                                                                const T* __restrict__ x, const T* __restrict__ z,
                                                                const float* __restrict__ bcoef, T* __restrict__ dx,
                                                                T* __restrict__ dres, int n, int d, int h, int w, int c, int act,
                                                                int mode, GnParamGrad pg) {
  if (pg.csum && blockIdx.x == 0) {  // pass 2b (gn_bwd_params_kernel) without a launch of its own
    for (int cc = threadIdx.x; cc < c; cc += 256) {
      double a = 0.0, b = 0.0;
      for (int i = 0; i < pg.n; ++i) {
        a += (double)pg.csum[((size_t)i * c + cc) * 2];
        b += (double)pg.csum[((size_t)i * c + cc) * 2 + 1];
      }
      if (pg.dbeta) pg.dbeta[cc] = (float)a;
      if (pg.dgamma) pg.dgamma[cc] = (float)b;
    }
  }
  const int od = d / 2, oh = h / 2, ow = w / 2, cv = c / VEC;
  const size_t total = (size_t)n * od * oh * ow * cv;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cc = (int)(i % cv);
  size_t v = i / cv;
  const int ox = (int)(v % ow);
  v /= ow;
  const int oy = (int)(v % oh);
  v /= oh;
  const int oz = (int)(v % od);
  const int nn = (int)(v / od);
  float k1[VEC], k2[VEC], k3[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const float* o = bcoef + ((size_t)nn * c + cc * VEC + k) * 3;
    k1[k] = o[0];
    k2[k] = o[1];
    k3[k] = o[2];
  }
  const F8 g = VecIO<T, VEC>::load(dyp, i * VEC);
  size_t src[8];
  F8 zv[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int iz = 2 * oz + (t >> 2), iy = 2 * oy + ((t >> 1) & 1), ix = 2 * ox + (t & 1);
    src[t] = ((((size_t)nn * d + iz) * h + iy) * w + ix) * c + (size_t)cc * VEC;
    zv[t] = VecIO<T, VEC>::load(z, src[t]);
  }
  int arg[VEC];
  float best[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    best[k] = -INFINITY;
    arg[k] = 0;
  }
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int k = 0; k < VEC; ++k)
      if (zv[t].v[k] > best[k] || zv[t].v[k] != zv[t].v[k]) {  // first maximum in scan order, NaN propagates (ATen)
        best[k] = zv[t].v[k];
        arg[k] = t;
      }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const F8 xv = VecIO<T, VEC>::load(x, src[t]);
    F8 g1;
#pragma unroll
    for (int k = 0; k < VEC; ++k) g1.v[k] = mode == MEDNET_POOL_MAX ? (arg[k] == t ? g.v[k] : 0.f) : 0.125f * g.v[k];
    if (add) {
      const F8 a = VecIO<T, VEC>::load(add, src[t]);
#pragma unroll
      for (int k = 0; k < VEC; ++k) g1.v[k] += a.v[k];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) g1.v[k] = (float)(T)g1.v[k];  // the value pool2_bwd_gn_kernel would have stored (and summed)
    act_grad_n<VEC>(g1.v, zv[t].v, act);
    F8 o;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      // plain fp32 FMAs on converted operands, THEN the rounding to T, as gn_bwd_apply_kernel has them: left alone the compiler
      // folds the conversions of the fp16 form into v_fma_mix_f32 / v_fma_mixlo_f16, whose results are not those of v_fma_f32 +
      // v_cvt (measured: about 4 % of the exact fp16 ties of dy3 then round the other way, a handful of elements per 10^6)
      float xf = xv.v[k];
      if (sizeof(T) == 2) asm volatile("" : "+v"(xf));
      o.v[k] = fmaf(k1[k], g1.v[k], fmaf(k2[k], xf, k3[k]));
      if (sizeof(T) == 2) asm volatile("" : "+v"(o.v[k]));
    }
    VecIO<T, VEC>::store(dx, src[t], o);
    VecIO<T, VEC>::store(dres, src[t], g1);
  }
}

// ---------------------------------------------------------------------------------------------- upsample + concat
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
  const float scale = (float)in / (float)out;  // ATen: compute_scales_value when only `size` is given
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}
// VEC channels per thread (8 when both channel counts allow 16-byte pieces): a piece never straddles the concat seam.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upcat_fwd_kernel(const T* __restrict__ enc, const T* __restrict__ x,
                                                        T* __restrict__ out, int n, int d, int h, int w, int ce, int xd,
                                                        int xh, int xw, int cx) {
  const int ct = ce + cx, pv = ct / VEC;
  const size_t total = (size_t)n * d * h * w * pv;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cc = (int)(i % pv) * VEC;
  size_t v = i / pv;
  if (cc < ce) {
    VecIO<T, VEC>::store(out, v * ct + cc, VecIO<T, VEC>::load(enc, v * ce + cc));
    return;
  }
  const int ox = (int)(v % w);
  size_t r = v / w;
  const int oy = (int)(r % h);
  r /= h;
  const int oz = (int)(r % d);
  const int nn = (int)(r / d);
  const int sz = nearest_src(oz, xd, d), sy = nearest_src(oy, xh, h), sx = nearest_src(ox, xw, w);
  VecIO<T, VEC>::store(out, v * ct + cc,
                       VecIO<T, VEC>::load(x, ((((size_t)nn * xd + sz) * xh + sy) * xw + sx) * cx + (cc - ce)));
}
// The same concatenation in the column-persistent layout of the GroupNorm kernels, with the statistics of what it writes: UNet3D's
// decoder opens with a GroupNorm over the concatenated tensor ('gcr', components.py:46-57 after :277-280), whose stand-alone
// statistics pass read all of it again (1.6 GB at 96 channels x 128^3 x 4).  A thread keeps its VEC channels -- from `enc` or from
// the low-resolution `x` -- walks the chunk's voxels four at a time, stores the pieces and sums them: partial[n][chunk][ct][2] =
// {sum v, sum v^2}, the layout of gn_partial_kernel (finalize: mednet_gn_finalize).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upcat_stats_fwd_kernel(const T* __restrict__ enc, const T* __restrict__ x, T* __restrict__ out,
                                                              float* __restrict__ partial, int d, int h, int w, int ce, int xd, int xh,
                                                              int xw, int cx, size_t chunk_vox) {
  __shared__ float lds[256 * 2 * VEC];
  const int ct = ce + cx;
  const Cols<VEC> L(ct);
  const int n = blockIdx.y;
  const size_t spatial = (size_t)d * h * w;
  const size_t v0 = (size_t)blockIdx.x * chunk_vox;
  const size_t v1 = v0 + chunk_vox < spatial ? v0 + chunk_vox : spatial;
  float acc[2 * VEC];
#pragma unroll
  for (int k = 0; k < 2 * VEC; ++k) acc[k] = 0.f;
  if (L.active) {
    const int cc = L.col * VEC;
    const bool from_enc = cc < ce;
    const T* src = from_enc ? enc + (size_t)n * spatial * ce + cc : x + (size_t)n * xd * xh * xw * cx + (cc - ce);
    T* dst = out + (size_t)n * spatial * ct + cc;
    auto src_index = [&](size_t v) -> size_t {
      if (from_enc) return v * ce;
      const int ox = (int)(v % w);
      const size_t r = v / w;
      const int oy = (int)(r % h), oz = (int)(r / h);
      return ((size_t)(nearest_src(oz, xd, d) * xh + nearest_src(oy, xh, h)) * xw + nearest_src(ox, xw, w)) * cx;
    };
    size_t v = v0 + L.row;
    for (; v + 3 * (size_t)L.rows < v1; v += 4 * (size_t)L.rows) {
      F8 xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[u] = VecIO<T, VEC>::load(src, src_index(v + (size_t)u * L.rows));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        VecIO<T, VEC>::store(dst, (v + (size_t)u * L.rows) * ct, xv[u]);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          acc[2 * k] += xv[u].v[k];
          acc[2 * k + 1] = fmaf(xv[u].v[k], xv[u].v[k], acc[2 * k + 1]);
        }
      }
    }
    for (; v < v1; v += L.rows) {
      const F8 xv = VecIO<T, VEC>::load(src, src_index(v));
      VecIO<T, VEC>::store(dst, v * ct, xv);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        acc[2 * k] += xv.v[k];
        acc[2 * k + 1] = fmaf(xv.v[k], xv.v[k], acc[2 * k + 1]);
      }
    }
  }
  float* rows = partial + ((size_t)n * gridDim.x + blockIdx.x) * ct * 2;
  column_reduce<2 * VEC>(acc, L.cols, L.rows, L.col, L.row, L.active, lds, rows);
}
// denc = dout[..., :ce]; dx[src] = sum over the destination voxels that map to src
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upcat_bwd_enc_kernel(const T* __restrict__ dout, T* __restrict__ denc,
                                                            size_t nvox, int ce, int ct) {
  const int pv = ce / VEC;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvox * pv) return;
  const size_t v = i / pv;
  const int cc = (int)(i % pv) * VEC;
  VecIO<T, VEC>::store(denc, v * ce + cc, VecIO<T, VEC>::load(dout, v * ct + cc));
}
__device__ __forceinline__ void dst_range(int src, int in, int out, int& lo, int& hi) {
  // all dst with nearest_src(dst) == src form a contiguous run; find it by scanning a small window
  int guess = (int)((double)src * out / in);
  lo = guess - 2 < 0 ? 0 : guess - 2;
  while (lo < out && nearest_src(lo, in, out) < src) ++lo;
  hi = lo;
  while (hi < out && nearest_src(hi, in, out) == src) ++hi;
}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upcat_bwd_x_kernel(const T* __restrict__ dout, T* __restrict__ dx, int n, int d,
                                                          int h, int w, int ce, int xd, int xh, int xw, int cx) {
  const int ct = ce + cx, pv = cx / VEC;
  const size_t total = (size_t)n * xd * xh * xw * pv;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cc = (int)(i % pv) * VEC;
  size_t v = i / pv;
  const size_t dst = v * cx + cc;
  const int sx = (int)(v % xw);
  v /= xw;
  const int sy = (int)(v % xh);
  v /= xh;
  const int sz = (int)(v % xd);
  const int nn = (int)(v / xd);
  int z0, z1, y0, y1, x0, x1;
  dst_range(sz, xd, d, z0, z1);
  dst_range(sy, xh, h, y0, y1);
  dst_range(sx, xw, w, x0, x1);
  F8 s;
#pragma unroll
  for (int k = 0; k < VEC; ++k) s.v[k] = 0.f;
  for (int oz = z0; oz < z1; ++oz)
    for (int oy = y0; oy < y1; ++oy)
      for (int ox = x0; ox < x1; ++ox) {
        const F8 g = VecIO<T, VEC>::load(dout, ((((size_t)nn * d + oz) * h + oy) * w + ox) * ct + ce + cc);
#pragma unroll
        for (int k = 0; k < VEC; ++k) s.v[k] += g.v[k];
      }
  VecIO<T, VEC>::store(dx, dst, s);
}

// ---------------------------------------------------------------------------------------------- Adam
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t count, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                   float gscale) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    float gr = g[i] * gscale;
    if (wd != 0.f) gr = fmaf(wd, p[i], gr);
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

// ---- fp16 storage: dynamic loss scaling without a host synchronisation (torch.cuda.amp.GradScaler's rule) ---------------
// state[8] (device, fp32): {scale, good steps since the last change, optimizer steps taken, found_inf, steps skipped, -, -, -}.  The loss is
// multiplied by state[0] on the device, so every gradient arrives times `scale`; grad_check raises found_inf if any
// gradient is not finite, adam_scaled divides by the scale and SKIPS the update when found_inf is set (bias correction
// uses the device-side step count, which a skipped step does not advance), scaler_update then halves the scale or
// counts a good step and doubles it every `growth_interval` of them.
__global__ __launch_bounds__(256) void grad_check_kernel(const float* __restrict__ g, size_t count, float* __restrict__ state) {
  const size_t stride = (size_t)gridDim.x * 256;
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    const float v = g[i];
    bad |= !(fabsf(v) <= 3.0e38f);  // NaN or +-inf
  }
  if (bad) state[3] = 1.f;  // (every writer stores the same value)
}
__global__ __launch_bounds__(256) void adam_scaled_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, size_t count, float lr,
                                                          float b1, float b2, float eps, float wd, float inv_world,
                                                          const float* __restrict__ state) {
  if (state[3] != 0.f) return;  // overflow somewhere in this step's gradients: no update
  const float t = state[2] + 1.f;
  const float bc1 = 1.f - powf(b1, t), bc2_sqrt = sqrtf(1.f - powf(b2, t));
  const float gscale = inv_world / state[0];
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    float gr = g[i] * gscale;
    if (wd != 0.f) gr = fmaf(wd, p[i], gr);
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}
__global__ void scaler_update_kernel(float* __restrict__ state, float growth, float backoff, float interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (state[3] != 0.f) {
    state[0] = fmaxf(state[0] * backoff, 1.f);  // never below 1: a persistent NaN (out-of-range labels) must not drive it to 0
    state[1] = 0.f;
    state[4] += 1.f;  // skipped steps, for the host to surface
  } else {
    state[2] += 1.f;
    state[1] += 1.f;
    if (state[1] >= interval) {
      state[0] *= growth;
      state[1] = 0.f;
    }
  }
  state[3] = 0.f;
}

}  // namespace mednet

// =================================================================================================== C ABI
using namespace mednet;

static inline unsigned flat_grid(size_t count, size_t per_block) {
  size_t b = (count + per_block - 1) / per_block;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}
static inline int pick_vec(int c, int dtype = MEDNET_BF16) {
  if (dtype == MEDNET_F32 && c % 4 == 0 && c / 4 <= 256 && tuning_option("gn_f32_vec4", 1)) return 4;
  return (c % 8 == 0 && c / 8 <= 256) ? 8 : 1;
}
// upper bound of the partial rows per sample any GroupNorm pass writes for C channels (chunks <= 1024), over both vector widths
static inline size_t gn_partial_rows_max(int c) {
  int best = 1;
  for (int dt = 0; dt < 2; ++dt) {
    const int vec = pick_vec(c, dt ? MEDNET_BF16 : MEDNET_F32), cols = c / vec;
    const int rpw = cols <= 256 ? lds_free_rows_per_wg(cols) : 0;
    if (rpw > best) best = rpw;
  }
  return (size_t)1024 * best;
}

extern "C" size_t mednet_gn_ws_bytes(int n, int c, size_t spatial) {
  // partial[n][rows][c][2] (rows = chunks, or 4 * chunks / (256 / cols) * chunks for the LDS-free backward pass)
  //   + bcoef[n][c][3] + csum[n][c][2]; the small arrays sit behind a fixed-size partial region
  (void)spatial;
  return ((size_t)n * gn_partial_rows_max(c) * c * 2 + (size_t)n * c * 5 + 64) * sizeof(float);
}

extern "C" int mednet_gn_stats(const void* x, const float* gamma, const float* beta, float* stats, float* coef,
                               int n, size_t spatial, int c, int groups, float eps, int dtype, void* ws,
                               size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(dtype_ok(dtype), MEDNET_E_DTYPE, "gn_stats: bad dtype %d", dtype);
  MEDNET_REQUIRE(n > 0 && c > 0 && groups > 0 && c % groups == 0 && spatial > 0, MEDNET_E_SHAPE,
                 "gn_stats: bad shape n=%d c=%d groups=%d", n, c, groups);
  const int vec = pick_vec(c, dtype);
  MEDNET_REQUIRE(c / vec <= 256, MEDNET_E_UNSUPPORTED, "gn_stats: C=%d unsupported (need C%%8==0 or C<=256)", c);
  MEDNET_REQUIRE(ws_bytes >= mednet_gn_ws_bytes(n, c, spatial), MEDNET_E_WORKSPACE, "gn_stats: workspace too small");
  size_t cv;
  unsigned chunks;
  chunk_plan(spatial, c, vec, cv, chunks);
  float* partial = (float*)ws;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(chunks, n);
#define GO(T, V) hipLaunchKernelGGL((gn_partial_kernel<T, V>), grid, dim3(256), 0, s, (const T*)x, partial, spatial, c, cv)
  if (dtype == MEDNET_F32) { if (vec == 4) GO(float, 4); else if (vec == 8) GO(float, 8); else GO(float, 1); }
  else if (dtype == MEDNET_BF16) { if (vec == 8) GO(bf16, 8); else GO(bf16, 1); }
  else { if (vec == 8) GO(f16, 8); else GO(f16, 1); }
#undef GO
  int rc = check_launch("gn_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(n * groups), dim3(chunks >= 4096 ? 1024 : 256), 0, s, partial, gamma, beta, stats, coef, c, groups,
                     (int)chunks, (double)spatial * (c / groups), eps);
  return check_launch("gn_finalize");
}

extern "C" int mednet_gn_finalize(const float* partial, int chunks, const float* gamma, const float* beta, float* stats,
                                  float* coef, int n, size_t spatial, int c, int groups, float eps, void* ws,
                                  size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(n > 0 && c > 0 && groups > 0 && c % groups == 0 && chunks > 0, MEDNET_E_SHAPE, "gn_finalize: bad shape");
  (void)ws;
  (void)ws_bytes;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(n * groups), dim3(chunks >= 4096 ? 1024 : 256), 0, s, partial, gamma, beta, stats, coef, c, groups, chunks,
                     (double)spatial * (c / groups), eps);
  return check_launch("gn_finalize");
}

extern "C" int mednet_gn_act_fwd(const void* x, const float* coef, const void* residual, void* z, int n,
                                 size_t spatial, int c, int act, int x_dtype, int z_dtype, mednet_stream stream) {
  MEDNET_REQUIRE(dtype_ok(x_dtype) && dtype_ok(z_dtype), MEDNET_E_DTYPE, "gn_act_fwd: bad dtype");
  MEDNET_REQUIRE(x_dtype == z_dtype, MEDNET_E_UNSUPPORTED, "gn_act_fwd: x and z must share a dtype");
  const int vec = pick_vec(c, x_dtype);
  MEDNET_REQUIRE(c / vec <= 256, MEDNET_E_UNSUPPORTED, "gn_act_fwd: C=%d unsupported", c);
  size_t cv;
  unsigned chunks;
  chunk_plan(spatial, c, vec, cv, chunks);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(chunks, n);
#define GO(T, V) hipLaunchKernelGGL((gn_act_fwd_kernel<T, T, V>), grid, dim3(256), 0, s, (const T*)x, coef, (const T*)residual, (T*)z, spatial, c, act, cv)
  if (x_dtype == MEDNET_F32) { if (vec == 4) GO(float, 4); else if (vec == 8) GO(float, 8); else GO(float, 1); }
  else if (x_dtype == MEDNET_BF16) { if (vec == 8) GO(bf16, 8); else GO(bf16, 1); }
  else { if (vec == 8) GO(f16, 8); else GO(f16, 1); }
#undef GO
  return check_launch("gn_act_fwd");
}

extern "C" int mednet_gn_act_bwd(const void* dz, const void* dz2, const void* x, const void* z, const float* coef,
                                 const float* stats, const float* gamma, void* dx, void* dres, float* dgamma,
                                 float* dbeta, int n,
                                 size_t spatial, int c, int groups, int act, int in_act, int dtype, void* ws, size_t ws_bytes,
                                 mednet_stream stream) {
  MEDNET_REQUIRE(dtype_ok(dtype), MEDNET_E_DTYPE, "gn_act_bwd: bad dtype");
  MEDNET_REQUIRE(c % groups == 0, MEDNET_E_SHAPE, "gn_act_bwd: C %% groups != 0");
  MEDNET_REQUIRE(act == MEDNET_ACT_NONE || z != nullptr || coef != nullptr, MEDNET_E_SHAPE,
                 "gn_act_bwd: act' needs the activated output z or the forward coefficients");
  const int vec = pick_vec(c, dtype);
  MEDNET_REQUIRE(c / vec <= 256, MEDNET_E_UNSUPPORTED, "gn_act_bwd: C=%d unsupported", c);
  MEDNET_REQUIRE(ws_bytes >= mednet_gn_ws_bytes(n, c, spatial), MEDNET_E_WORKSPACE, "gn_act_bwd: workspace too small");
  size_t cv;
  unsigned chunks;
  chunk_plan(spatial, c, vec, cv, chunks);
  float* partial = (float*)ws;
  float* bcoef = partial + (size_t)n * gn_partial_rows_max(c) * c * 2;
  float* csum = bcoef + (size_t)n * c * 3;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(chunks, n);
  const int rpw = tuning_option("gn_lds_free", 1) ? lds_free_rows_per_wg(c / vec) : 0;
#define GO(T, V) hipLaunchKernelGGL((gn_bwd_partial_kernel<T, V>), grid, dim3(256), 0, s, (const T*)dz, (const T*)dz2, (const T*)x, (const T*)z, coef, stats, partial, spatial, c, groups, act, cv, rpw)
  if (dtype == MEDNET_F32) { if (vec == 4) GO(float, 4); else if (vec == 8) GO(float, 8); else GO(float, 1); }
  else if (dtype == MEDNET_BF16) { if (vec == 8) GO(bf16, 8); else GO(bf16, 1); }
  else { if (vec == 8) GO(f16, 8); else GO(f16, 1); }
#undef GO
  int rc = check_launch("gn_bwd_partial");
  if (rc) return rc;
  GnParamGrad pg;
  if (gn_bwd_one_launch(c, groups)) {
    hipLaunchKernelGGL(gn_bwd_reduce_finalize_kernel<false>, dim3(n * groups), dim3(256), 0, s, partial, stats, gamma, csum, bcoef, c,
                       groups, (int)chunks * (rpw > 0 ? rpw : 1), (double)spatial * (c / groups));
    rc = check_launch("gn_bwd_reduce_finalize");
    if (rc) return rc;
    if (dgamma || dbeta) pg = GnParamGrad{csum, dgamma, dbeta, n};
  } else {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(n * c), dim3(64), 0, s, partial, csum, c, (int)chunks * (rpw > 0 ? rpw : 1));
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(n * groups), dim3(64), 0, s, csum, stats, gamma, bcoef, c, groups,
                       (double)spatial * (c / groups));
    rc = check_launch("gn_bwd_finalize");
    if (rc) return rc;
    if (dgamma || dbeta) {
      hipLaunchKernelGGL(gn_bwd_params_kernel, dim3((c + 255) / 256), dim3(256), 0, s, csum, dgamma, dbeta, n, c);
      rc = check_launch("gn_bwd_params");
      if (rc) return rc;
    }
  }
#define GO(T, V) hipLaunchKernelGGL((gn_bwd_apply_kernel<T, V>), grid, dim3(256), 0, s, (const T*)dz, (const T*)dz2, (const T*)x, (const T*)z, coef, bcoef, (T*)dx, (T*)dres, spatial, c, act, cv, in_act, pg)
  if (dtype == MEDNET_F32) { if (vec == 4) GO(float, 4); else if (vec == 8) GO(float, 8); else GO(float, 1); }
  else if (dtype == MEDNET_BF16) { if (vec == 8) GO(bf16, 8); else GO(bf16, 1); }
  else { if (vec == 8) GO(f16, 8); else GO(f16, 1); }
#undef GO
  return check_launch("gn_bwd_apply");
}

static int gn_act_bwd_fused_impl(const void* dz, const void* x, const void* z, const float* coef, const float* stats,
                                 const float* gamma, const float* fused_partial, int rows, void* dx, void* dres,
                                 float* dgamma, float* dbeta, int n, size_t spatial, int c, int groups, int act, int in_act,
                                 int dtype, void* ws, size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(dtype_ok(dtype), MEDNET_E_DTYPE, "gn_act_bwd_fused: bad dtype");
  MEDNET_REQUIRE(c % groups == 0 && rows > 0 && fused_partial && coef, MEDNET_E_SHAPE, "gn_act_bwd_fused: bad arguments");
  const int vec = pick_vec(c, dtype);
  MEDNET_REQUIRE(c / vec <= 256, MEDNET_E_UNSUPPORTED, "gn_act_bwd_fused: C=%d unsupported", c);
  MEDNET_REQUIRE(ws_bytes >= mednet_gn_ws_bytes(n, c, spatial), MEDNET_E_WORKSPACE, "gn_act_bwd_fused: workspace too small");
  size_t cv;
  unsigned chunks;
  chunk_plan(spatial, c, vec, cv, chunks);
  float* bcoef = (float*)ws + (size_t)n * gn_partial_rows_max(c) * c * 2;
  float* csum = bcoef + (size_t)n * c * 3;
  hipStream_t s = (hipStream_t)stream;
  const int skip = tuning_option("gn_bwd_skip", 0);  // timing probes only (tools/probes/stall_probe.py): bit k drops sub-kernel k
  GnParamGrad pg;
  int rc;
  if (gn_bwd_one_launch(c, groups) && !skip) {
    hipLaunchKernelGGL(gn_bwd_reduce_finalize_kernel<true>, dim3(n * groups), dim3(256), 0, s, fused_partial, stats, gamma, csum, bcoef,
                       c, groups, rows, (double)spatial * (c / groups));
    rc = check_launch("gn_bwd_reduce_finalize");
    if (rc) return rc;
    if (dgamma || dbeta) pg = GnParamGrad{csum, dgamma, dbeta, n};
  } else {
    if (!(skip & 1))
      hipLaunchKernelGGL(reduce_partials_dux_kernel, dim3(n * c), dim3(64), 0, s, fused_partial, stats, csum, c, groups, rows);
    if (!(skip & 2))
      hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(n * groups), dim3(64), 0, s, csum, stats, gamma, bcoef, c, groups,
                         (double)spatial * (c / groups));
    rc = check_launch("gn_bwd_finalize");
    if (rc) return rc;
    if (skip & 8) return MEDNET_OK;
    if ((dgamma || dbeta) && !(skip & 4)) {
      hipLaunchKernelGGL(gn_bwd_params_kernel, dim3((c + 255) / 256), dim3(256), 0, s, csum, dgamma, dbeta, n, c);
      rc = check_launch("gn_bwd_params");
      if (rc) return rc;
    }
  }
  const dim3 grid(chunks, n);
#define GO(T, V) hipLaunchKernelGGL((gn_bwd_apply_kernel<T, V>), grid, dim3(256), 0, s, (const T*)dz, (const T*)nullptr, (const T*)x, (const T*)z, coef, bcoef, (T*)dx, (T*)dres, spatial, c, act, cv, in_act, pg)
  if (dtype == MEDNET_F32) { if (vec == 4) GO(float, 4); else if (vec == 8) GO(float, 8); else GO(float, 1); }
  else if (dtype == MEDNET_BF16) { if (vec == 8) GO(bf16, 8); else GO(bf16, 1); }
  else { if (vec == 8) GO(f16, 8); else GO(f16, 1); }
#undef GO
  return check_launch("gn_bwd_apply");
}

extern "C" int mednet_gn_act_bwd_fused(const void* dz, const void* x, const float* coef, const float* stats,
                                       const float* gamma, const float* fused_partial, int rows, void* dx, float* dgamma,
                                       float* dbeta, int n, size_t spatial, int c, int groups, int act, int in_act, int dtype,
                                       void* ws, size_t ws_bytes, mednet_stream stream) {
  return gn_act_bwd_fused_impl(dz, x, nullptr, coef, stats, gamma, fused_partial, rows, dx, nullptr, dgamma, dbeta, n, spatial, c,
                               groups, act, in_act, dtype, ws, ws_bytes, stream);
}
// Passes 1b + 2 of mednet_gn_act_bwd_fused WITHOUT its apply pass: the per-(sample, channel) coefficients {k1, k2, k3} of
//   dy = k1 * du + k2 * y + k3        (du = dz * act'(coef_a * y + coef_b))
// go to `bcoef` ([n][c][3] floats) for a consumer that applies them while it reads (dz, y) for its own purpose
// (mednet_conv3d_wgrad_c1_gn: the first layer's weight gradient, whose dy is never stored), and the parameter gradients to
// dgamma / dbeta.  Same kernels, same order of sums as the fused form => the same coefficient bits.
extern "C" int mednet_gn_bwd_coefficients(const float* stats, const float* gamma, const float* fused_partial, int rows,
                                          float* bcoef, float* dgamma, float* dbeta, int n, size_t spatial, int c, int groups,
                                          void* ws, size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(c % groups == 0 && rows > 0 && fused_partial && bcoef && stats && gamma, MEDNET_E_SHAPE,
                 "gn_bwd_coefficients: bad arguments");
  MEDNET_REQUIRE(ws_bytes >= mednet_gn_ws_bytes(n, c, spatial), MEDNET_E_WORKSPACE, "gn_bwd_coefficients: workspace too small");
  float* csum = (float*)ws;  // [n][c][2]
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (gn_bwd_one_launch(c, groups)) {
    hipLaunchKernelGGL(gn_bwd_reduce_finalize_kernel<true>, dim3(n * groups), dim3(256), 0, s, fused_partial, stats, gamma, csum, bcoef,
                       c, groups, rows, (double)spatial * (c / groups));
    rc = check_launch("gn_bwd_reduce_finalize");
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(reduce_partials_dux_kernel, dim3(n * c), dim3(64), 0, s, fused_partial, stats, csum, c, groups, rows);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(n * groups), dim3(64), 0, s, csum, stats, gamma, bcoef, c, groups,
                       (double)spatial * (c / groups));
    rc = check_launch("gn_bwd_finalize");
    if (rc) return rc;
  }
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(gn_bwd_params_kernel, dim3((c + 255) / 256), dim3(256), 0, s, csum, dgamma, dbeta, n, c);
    rc = check_launch("gn_bwd_params");
    if (rc) return rc;
  }
  return MEDNET_OK;
}

// ... for the residual layer of an ExtResNetBlock: the activation derivative comes from the block OUTPUT z, and du is also
// the gradient of the residual branch (dres)
extern "C" int mednet_gn_act_bwd_fused_res(const void* dz, const void* x, const void* z, const float* coef, const float* stats,
                                           const float* gamma, const float* fused_partial, int rows, void* dx, void* dres,
                                           float* dgamma, float* dbeta, int n, size_t spatial, int c, int groups, int act,
                                           int dtype, void* ws, size_t ws_bytes, mednet_stream stream) {
  MEDNET_REQUIRE(z && dres, MEDNET_E_SHAPE, "gn_act_bwd_fused_res: z and dres are required");
  return gn_act_bwd_fused_impl(dz, x, z, coef, stats, gamma, fused_partial, rows, dx, dres, dgamma, dbeta, n, spatial, c, groups,
                               act, MEDNET_ACT_NONE, dtype, ws, ws_bytes, stream);
}

// ... for the residual layer of an ENCODER block whose output gradient dz = pooling backward(dy_pool) + skip gradient was not
// materialised (mednet_pool2_bwd_gn with dx = NULL took the sums): the apply pass rebuilds dz per pooling window.
extern "C" int mednet_gn_act_bwd_fused_res_pool(const void* dy_pool, const void* skip_grad, const void* x, const void* z,
                                                const float* stats, const float* gamma, const float* fused_partial, int rows,
                                                void* dx, void* dres, float* dgamma, float* dbeta, int n, int d, int h, int w, int c,
                                                int groups, int act, int pool_mode, int dtype, void* ws, size_t ws_bytes,
                                                mednet_stream stream) {
  MEDNET_REQUIRE(dtype_ok(dtype), MEDNET_E_DTYPE, "gn_act_bwd_fused_res_pool: bad dtype");
  MEDNET_REQUIRE(c % groups == 0 && rows > 0 && fused_partial && dy_pool && x && z && dx && dres, MEDNET_E_SHAPE,
                 "gn_act_bwd_fused_res_pool: bad arguments");
  MEDNET_REQUIRE(d % 2 == 0 && h % 2 == 0 && w % 2 == 0 && c % 8 == 0, MEDNET_E_UNSUPPORTED,
                 "gn_act_bwd_fused_res_pool: even extents and C %% 8 == 0 (%dx%dx%d, C=%d)", d, h, w, c);
  const size_t spatial = (size_t)d * h * w;
  MEDNET_REQUIRE(ws_bytes >= mednet_gn_ws_bytes(n, c, spatial), MEDNET_E_WORKSPACE, "gn_act_bwd_fused_res_pool: workspace too small");
  float* bcoef = (float*)ws + (size_t)n * gn_partial_rows_max(c) * c * 2;
  float* csum = bcoef + (size_t)n * c * 3;
  hipStream_t s = (hipStream_t)stream;
  GnParamGrad pg;
  int rc;
  if (gn_bwd_one_launch(c, groups)) {
    hipLaunchKernelGGL(gn_bwd_reduce_finalize_kernel<true>, dim3(n * groups), dim3(256), 0, s, fused_partial, stats, gamma, csum, bcoef,
                       c, groups, rows, (double)spatial * (c / groups));
    rc = check_launch("gn_bwd_reduce_finalize");
    if (rc) return rc;
    if (dgamma || dbeta) pg = GnParamGrad{csum, dgamma, dbeta, n};
  } else {
    hipLaunchKernelGGL(reduce_partials_dux_kernel, dim3(n * c), dim3(64), 0, s, fused_partial, stats, csum, c, groups, rows);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(n * groups), dim3(64), 0, s, csum, stats, gamma, bcoef, c, groups,
                       (double)spatial * (c / groups));
    rc = check_launch("gn_bwd_finalize");
    if (rc) return rc;
    if (dgamma || dbeta) {
      hipLaunchKernelGGL(gn_bwd_params_kernel, dim3((c + 255) / 256), dim3(256), 0, s, csum, dgamma, dbeta, n, c);
      rc = check_launch("gn_bwd_params");
      if (rc) return rc;
    }
  }
  const int vec = dtype == MEDNET_F32 ? 4 : 8;
  const size_t total = (size_t)n * (d / 2) * (h / 2) * (w / 2) * (c / vec);
  const dim3 grid((unsigned)((total + 255) / 256));
#define GO(T, V) hipLaunchKernelGGL((gn_bwd_apply_pool_kernel<T, V>), grid, dim3(256), 0, s, (const T*)dy_pool, (const T*)skip_grad, (const T*)x, (const T*)z, bcoef, (T*)dx, (T*)dres, n, d, h, w, c, act, pool_mode, pg)
  if (dtype == MEDNET_F32) GO(float, 4);
  else if (dtype == MEDNET_BF16) GO(bf16, 8);
  else GO(f16, 8);
#undef GO
  return check_launch("gn_bwd_apply_pool");
}

extern "C" int mednet_act_fwd(const void* x, void* z, size_t count, int act, int dtype, mednet_stream stream) {
  MEDNET_REQUIRE(dtype_ok(dtype), MEDNET_E_DTYPE, "act_fwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  const unsigned g = flat_grid(count, 2048);
  if (dtype == MEDNET_F32) hipLaunchKernelGGL(act_fwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, (float*)z, count, act);
  else if (dtype == MEDNET_BF16) hipLaunchKernelGGL(act_fwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)z, count, act);
  else hipLaunchKernelGGL(act_fwd_kernel<f16>, dim3(g), dim3(256), 0, s, (const f16*)x, (f16*)z, count, act);
  return check_launch("act_fwd");
}
extern "C" int mednet_act_bwd(const void* dz, const void* z, void* dx, size_t count, int act, int dtype,
                              mednet_stream stream) {
  MEDNET_REQUIRE(dtype_ok(dtype), MEDNET_E_DTYPE, "act_bwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  const unsigned g = flat_grid(count, 2048);
  if (dtype == MEDNET_F32) hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)dz, (const float*)z, (float*)dx, count, act);
  else if (dtype == MEDNET_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)dz, (const bf16*)z, (bf16*)dx, count, act);
  else hipLaunchKernelGGL(act_bwd_kernel<f16>, dim3(g), dim3(256), 0, s, (const f16*)dz, (const f16*)z, (f16*)dx, count, act);
  return check_launch("act_bwd");
}
extern "C" int mednet_add(const void* a, const void* b, void* out, size_t count, int dtype, mednet_stream stream) {
  MEDNET_REQUIRE(dtype_ok(dtype), MEDNET_E_DTYPE, "add: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  const unsigned g = flat_grid(count, 2048);
  if (dtype == MEDNET_F32) hipLaunchKernelGGL(add_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)out, count);
  else if (dtype == MEDNET_BF16) hipLaunchKernelGGL(add_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, (bf16*)out, count);
  else hipLaunchKernelGGL(add_kernel<f16>, dim3(g), dim3(256), 0, s, (const f16*)a, (const f16*)b, (f16*)out, count);
  return check_launch("add");
}

extern "C" int mednet_pool2_fwd(const void* x, void* y, int n, int d, int h, int w, int c, int mode, int dtype,
                                mednet_stream stream) {
  MEDNET_REQUIRE(dtype_ok(dtype), MEDNET_E_DTYPE, "pool2_fwd: bad dtype");
  MEDNET_REQUIRE(d >= 2 && h >= 2 && w >= 2, MEDNET_E_SHAPE, "pool2_fwd: dims must be >= 2");
  const int vec = c % 8 == 0 ? 8 : 1;
  const size_t total = (size_t)n * (d / 2) * (h / 2) * (w / 2) * (c / vec);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((total + 255) / 256));
#define GO(T, V) hipLaunchKernelGGL((pool2_fwd_kernel<T, V>), grid, dim3(256), 0, s, (const T*)x, (T*)y, n, d, h, w, c, mode)
  if (dtype == MEDNET_F32) { if (vec == 4) GO(float, 4); else if (vec == 8) GO(float, 8); else GO(float, 1); }
  else if (dtype == MEDNET_BF16) { if (vec == 8) GO(bf16, 8); else GO(bf16, 1); }
  else { if (vec == 8) GO(f16, 8); else GO(f16, 1); }
#undef GO
  return check_launch("pool2_fwd");
}
extern "C" int mednet_gn_act_pool_supported(int d, int h, int w, int c, int dtype) {
  return d >= 2 && h >= 2 && w >= 2 && d % 2 == 0 && h % 2 == 0 && w % 2 == 0 && c % 8 == 0 && dtype_ok(dtype) && tuning_option("gn_pool_fuse", 1);
}
extern "C" int mednet_gn_act_pool_fwd(const void* x, const float* coef, const void* residual, void* z, void* pooled, int n, int d,
                                      int h, int w, int c, int act, int mode, int dtype, mednet_stream stream) {
  MEDNET_REQUIRE(mednet_gn_act_pool_supported(d, h, w, c, dtype), MEDNET_E_UNSUPPORTED,
                 "gn_act_pool_fwd: even dims and C %% 8 == 0 only (%dx%dx%d, C=%d, dtype %d)", d, h, w, c, dtype);
  MEDNET_REQUIRE(n > 0 && x && coef && z && pooled, MEDNET_E_SHAPE, "gn_act_pool_fwd: bad arguments");
  const int vec = dtype == MEDNET_F32 ? 4 : 8;  // (fp32: one 16-byte access per lane and tensor, as the other fp32 GroupNorm kernels)
  const size_t total = (size_t)n * (d / 2) * (h / 2) * (w / 2) * (c / vec);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == MEDNET_BF16)
    hipLaunchKernelGGL((gn_act_pool_fwd_kernel<bf16, 8>), grid, dim3(256), 0, s, (const bf16*)x, coef, (const bf16*)residual, (bf16*)z, (bf16*)pooled, n, d, h, w, c, act, mode);
  else if (dtype == MEDNET_F16)
    hipLaunchKernelGGL((gn_act_pool_fwd_kernel<f16, 8>), grid, dim3(256), 0, s, (const f16*)x, coef, (const f16*)residual, (f16*)z, (f16*)pooled, n, d, h, w, c, act, mode);
  else
    hipLaunchKernelGGL((gn_act_pool_fwd_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)x, coef, (const float*)residual, (float*)z, (float*)pooled, n, d, h, w, c, act, mode);
  return check_launch("gn_act_pool_fwd");
}

extern "C" int mednet_pool2_bwd_act(const void* dy, const void* x, const void* add, int add_channels, void* dx, int n, int d, int h,
                                    int w, int c, int mode, int in_act, int dtype, mednet_stream stream) {
  MEDNET_REQUIRE(dtype_ok(dtype), MEDNET_E_DTYPE, "pool2_bwd: bad dtype");
  if (add_channels <= 0) add_channels = c;
  MEDNET_REQUIRE(add_channels == c || (add && add_channels > c && add_channels % 8 == 0 && c % 8 == 0 && !((d | h | w) & 1)), MEDNET_E_UNSUPPORTED,
                 "pool2_bwd_act: a strided second gradient needs channel counts that are multiples of 8 and even extents");
  MEDNET_REQUIRE(in_act == MEDNET_ACT_NONE || !((d | h | w) & 1), MEDNET_E_UNSUPPORTED,
                 "pool2_bwd_act: the activation derivative is folded in for even extents only (%dx%dx%d)", d, h, w);
  hipStream_t s = (hipStream_t)stream;
  if ((d | h | w) & 1) {  // odd tails are never pooled: their gradient is zero (or just `add`)
    const size_t bytes = (size_t)n * d * h * w * c * dtype_size(dtype);
    const hipError_t e = add ? hipMemcpyAsync(dx, add, bytes, hipMemcpyDeviceToDevice, s) : hipMemsetAsync(dx, 0, bytes, s);
    if (e != hipSuccess) return fail(MEDNET_E_HIP, "pool2_bwd: tail fill failed");
  }
  const int vec = c % 8 == 0 ? 8 : 1;
  const size_t total = (size_t)n * (d / 2) * (h / 2) * (w / 2) * (c / vec);
  const dim3 grid((unsigned)((total + 255) / 256));
#define GO(T, V) hipLaunchKernelGGL((pool2_bwd_kernel<T, V>), grid, dim3(256), 0, s, (const T*)dy, (const T*)x, (const T*)add, (T*)dx, n, d, h, w, c, mode, in_act, add_channels)
  if (dtype == MEDNET_F32) { if (vec == 4) GO(float, 4); else if (vec == 8) GO(float, 8); else GO(float, 1); }
  else if (dtype == MEDNET_BF16) { if (vec == 8) GO(bf16, 8); else GO(bf16, 1); }
  else { if (vec == 8) GO(f16, 8); else GO(f16, 1); }
#undef GO
  return check_launch("pool2_bwd");
}
extern "C" int mednet_pool2_bwd(const void* dy, const void* x, const void* add, void* dx, int n, int d, int h, int w,
                                int c, int mode, int dtype, mednet_stream stream) {
  return mednet_pool2_bwd_act(dy, x, add, 0, dx, n, d, h, w, c, mode, MEDNET_ACT_NONE, dtype, stream);
}

static bool pool2_gn_ok(int d, int h, int w, int c, int dtype) {
  const int cv = c / 8;
  return c % 8 == 0 && cv <= 64 && (cv & (cv - 1)) == 0 && !((d | h | w) & 1) && dtype_ok(dtype);  // (fp32 storage too, round 3)
}
extern "C" int mednet_pool2_bwd_gn_rows(int n, int d, int h, int w, int c, int dtype) {
  (void)n;
  if (!pool2_gn_ok(d, h, w, c, dtype) || !tuning_option("gn3_fuse", 1)) return 0;
  const size_t per = (size_t)(d / 2) * (h / 2) * (w / 2) * (c / 8);
  return 4 * (int)((per + 256 * POOL_GN_PV - 1) / (256 * POOL_GN_PV));
}
extern "C" int mednet_pool2_bwd_gn(const void* dy, const void* x, const void* add, void* dx, const void* gn_y, int gn_act,
                                   float* gn_partial, int n, int d, int h, int w, int c, int mode, int dtype,
                                   mednet_stream stream) {
  const int rows = mednet_pool2_bwd_gn_rows(n, d, h, w, c, dtype);
  MEDNET_REQUIRE(rows > 0, MEDNET_E_UNSUPPORTED, "pool2_bwd_gn: shape %dx%dx%d c=%d dtype=%d not supported", d, h, w, c, dtype);
  MEDNET_REQUIRE(gn_y && gn_partial, MEDNET_E_SHAPE, "pool2_bwd_gn: gn_y and gn_partial are required");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(rows / 4), n);
  if (dtype == MEDNET_F32)
    hipLaunchKernelGGL(pool2_bwd_gn_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, (const float*)x, (const float*)add,
                       (float*)dx, (const float*)gn_y, gn_partial, d, h, w, c, mode, gn_act);
  else if (dtype == MEDNET_BF16)
    hipLaunchKernelGGL(pool2_bwd_gn_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (const bf16*)add, (bf16*)dx,
                       (const bf16*)gn_y, gn_partial, d, h, w, c, mode, gn_act);
  else
    hipLaunchKernelGGL(pool2_bwd_gn_kernel<f16>, grid, dim3(256), 0, s, (const f16*)dy, (const f16*)x, (const f16*)add, (f16*)dx,
                       (const f16*)gn_y, gn_partial, d, h, w, c, mode, gn_act);
  return check_launch("pool2_bwd_gn");
}

extern "C" int mednet_upcat_fwd(const void* enc, const void* x, void* out, int n, int d, int h, int w, int c_enc,
                                int xd, int xh, int xw, int c_x, int dtype, mednet_stream stream) {
  MEDNET_REQUIRE(dtype_ok(dtype), MEDNET_E_DTYPE, "upcat_fwd: bad dtype");
  const int vec = (c_enc % 8 == 0 && c_x % 8 == 0) ? 8 : 1;
  const size_t total = (size_t)n * d * h * w * ((c_enc + c_x) / vec);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((total + 255) / 256));
#define GO(T, V) hipLaunchKernelGGL((upcat_fwd_kernel<T, V>), grid, dim3(256), 0, s, (const T*)enc, (const T*)x, (T*)out, n, d, h, w, c_enc, xd, xh, xw, c_x)
  if (dtype == MEDNET_F32) { if (vec == 4) GO(float, 4); else if (vec == 8) GO(float, 8); else GO(float, 1); }
  else if (dtype == MEDNET_BF16) { if (vec == 8) GO(bf16, 8); else GO(bf16, 1); }
  else { if (vec == 8) GO(f16, 8); else GO(f16, 1); }
#undef GO
  return check_launch("upcat_fwd");
}
// ... with the GroupNorm partial sums of the concatenated tensor: partial[n][mednet_upcat_stats_chunks][c_enc + c_x][2]
extern "C" int mednet_upcat_stats_chunks(int n, int d, int h, int w, int c_enc, int c_x, int dtype) {
  (void)n;
  const int ct = c_enc + c_x;
  if (!dtype_ok(dtype) || c_enc % 8 || c_x % 8 || ct / 8 > 256 || !tuning_option("upcat_stats", 1)) return 0;
  size_t cv;
  unsigned chunks;
  chunk_plan((size_t)d * h * w, ct, 8, cv, chunks);
  return (int)chunks;
}
extern "C" int mednet_upcat_fwd_stats(const void* enc, const void* x, void* out, float* partial, int n, int d, int h, int w,
                                      int c_enc, int xd, int xh, int xw, int c_x, int dtype, mednet_stream stream) {
  MEDNET_REQUIRE(mednet_upcat_stats_chunks(n, d, h, w, c_enc, c_x, dtype) > 0 && partial, MEDNET_E_UNSUPPORTED,
                 "upcat_fwd_stats: %d + %d channels, dtype %d", c_enc, c_x, dtype);
  size_t cv;
  unsigned chunks;
  chunk_plan((size_t)d * h * w, c_enc + c_x, 8, cv, chunks);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(chunks, n);
#define GO(T) hipLaunchKernelGGL((upcat_stats_fwd_kernel<T, 8>), grid, dim3(256), 0, s, (const T*)enc, (const T*)x, (T*)out, partial, d, h, w, c_enc, xd, xh, xw, c_x, cv)
  if (dtype == MEDNET_F32) GO(float);
  else if (dtype == MEDNET_BF16) GO(bf16);
  else GO(f16);
#undef GO
  return check_launch("upcat_fwd_stats");
}
extern "C" int mednet_upcat_bwd(const void* dout, void* denc, void* dx, int n, int d, int h, int w, int c_enc, int xd,
                                int xh, int xw, int c_x, int dtype, mednet_stream stream) {
  MEDNET_REQUIRE(dtype_ok(dtype), MEDNET_E_DTYPE, "upcat_bwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  const size_t nvox = (size_t)n * d * h * w;
  const int ct = c_enc + c_x;
  const int vec = (c_enc % 8 == 0 && c_x % 8 == 0) ? 8 : 1;
  const dim3 g1((unsigned)((nvox * (c_enc / vec) + 255) / 256));
  const size_t tx = (size_t)n * xd * xh * xw * (c_x / vec);
  const dim3 g2((unsigned)((tx + 255) / 256));
#define GO(T, V)                                                                                                        \
  do {                                                                                                                  \
    if (denc) hipLaunchKernelGGL((upcat_bwd_enc_kernel<T, V>), g1, dim3(256), 0, s, (const T*)dout, (T*)denc, nvox, c_enc, ct); \
    hipLaunchKernelGGL((upcat_bwd_x_kernel<T, V>), g2, dim3(256), 0, s, (const T*)dout, (T*)dx, n, d, h, w, c_enc, xd,  \
                       xh, xw, c_x);                                                                                    \
  } while (0)
  if (dtype == MEDNET_F32) { if (vec == 4) GO(float, 4); else if (vec == 8) GO(float, 8); else GO(float, 1); }
  else if (dtype == MEDNET_BF16) { if (vec == 8) GO(bf16, 8); else GO(bf16, 1); }
  else { if (vec == 8) GO(f16, 8); else GO(f16, 1); }
#undef GO
  return check_launch("upcat_bwd");
}

extern "C" int mednet_adam_step_scaled(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1,
                                       float beta2, float eps, float weight_decay, float inv_world, float* scaler_state,
                                       float growth_factor, float backoff_factor, int growth_interval, mednet_stream stream) {
  MEDNET_REQUIRE(scaler_state != nullptr && growth_interval >= 1, MEDNET_E_SHAPE, "adam_step_scaled: scaler state required");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(grad_check_kernel, dim3(flat_grid(count, 1024)), dim3(256), 0, s, g, count, scaler_state);
  hipLaunchKernelGGL(adam_scaled_kernel, dim3(flat_grid(count, 1024)), dim3(256), 0, s, p, g, m, v, count, lr, beta1, beta2, eps,
                     weight_decay, inv_world, scaler_state);
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, s, scaler_state, growth_factor, backoff_factor,
                     (float)growth_interval);
  return check_launch("adam_step_scaled");
}

extern "C" int mednet_adam_step(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, float grad_scale,
                                mednet_stream stream) {
  MEDNET_REQUIRE(step >= 1, MEDNET_E_SHAPE, "adam_step: step must be >= 1");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(flat_grid(count, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, count, lr,
                     beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
  return check_launch("adam_step");
}
