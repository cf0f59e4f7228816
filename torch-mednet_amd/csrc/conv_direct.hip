// Direct (VALU, fp32-accumulate) convolution family: the shape-generic path of libmednet_hip.
//
// One template covers the three index maps the U-Net needs,
//     out[vo][m] = bias[m] + sum_t sum_k in[map(vo, t)][k] * P[t][k][m]
//   MAP_CONV   : in = vo + t - pad                 nn.Conv3d fwd, and its data gradient with the flipped pack
//   MAP_CT_FWD : in = (vo + 1 - t) / 2 if even     nn.ConvTranspose3d(k3,s2,p1,op1) forward (gather form)
//   MAP_CT_DG  : in = 2*vo - 1 + t                 ... and its data gradient
// (reference call sites: components.py:8-9,44 ; :259-264 ; model.py:77,179).
// A lane owns one output voxel and COB output channels; the weight index (t,k,m) is wave-uniform, so the
// compiler keeps weights on the scalar path (s_load + v_fmac with an SGPR operand) and the vector memory path
// only carries activations, read 8 channels (16-32 B) at a time from the channels-last row of the voxel.
// It serves Cin=1 (first layer), the 1x1x1 head, ConvTranspose and every channel count the MFMA kernels do not
// take, in both storage precisions.
#include "conv.h"

namespace mednet_f16 {  // the fp16 build of conv_mfma.hip (first-layer weight gradient on the matrix cores)
#include "conv_mfma_decl.inc"
}

namespace mednet {

template <int MAP>
__device__ __forceinline__ bool map_coord(int o, int t, int ks, int in_extent, int& i) {
  if (MAP == MAP_CONV) {
    i = o + t - (ks >> 1);
  } else if (MAP == MAP_CT_FWD) {
    const int u = o + 1 - t;
    if (u & 1) return false;
    i = u >> 1;  // u >= -1; u=-1 is odd -> rejected above
  } else {
    i = 2 * o - 1 + t;
  }
  return i >= 0 && i < in_extent;
}

template <typename TI, typename TO, int MAP, int COB>
__global__ __launch_bounds__(256) void conv_direct_kernel(const TI* __restrict__ x, const float* __restrict__ P,
                                                          const float* __restrict__ bias,
                                                          const TO* __restrict__ skip, TO* __restrict__ y,
                                                          ConvGeom g) {
  const size_t ovol = (size_t)g.od * g.oh * g.ow;
  const size_t ivol = (size_t)g.id * g.ih * g.iw;
  const size_t nvox = (size_t)g.n * ovol;
  const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= nvox) return;
  const int n = (int)(v / ovol);
  const size_t sp = v - (size_t)n * ovol;
  const int oz = (int)(sp / ((size_t)g.oh * g.ow));
  const int rem = (int)(sp - (size_t)oz * g.oh * g.ow);
  const int oy = rem / g.ow;
  const int ox = rem - oy * g.ow;
  const int m0 = blockIdx.y * COB;

  float acc[COB];
#pragma unroll
  for (int j = 0; j < COB; ++j) acc[j] = bias ? bias[m0 + j] : 0.f;

  const bool vec8 = (!g.in_planar) && (g.k % 8 == 0);
  for (int tz = 0; tz < g.ks; ++tz) {
    int iz;
    if (!map_coord<MAP>(oz, tz, g.ks, g.id, iz)) continue;
    for (int ty = 0; ty < g.ks; ++ty) {
      int iy;
      if (!map_coord<MAP>(oy, ty, g.ks, g.ih, iy)) continue;
      for (int tx = 0; tx < g.ks; ++tx) {
        int ix;
        if (!map_coord<MAP>(ox, tx, g.ks, g.iw, ix)) continue;
        const int t = (tz * g.ks + ty) * g.ks + tx;
        const size_t isp = ((size_t)iz * g.ih + iy) * g.iw + ix;
        const float* Pt = P + (size_t)t * g.k * g.m + m0;
        if (vec8) {
          const TI* xp = x + ((size_t)n * ivol + isp) * g.k;
          for (int k = 0; k < g.k; k += 8) {
            const F8 xv = ld8(xp, k);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const float* Pk = Pt + (size_t)(k + c) * g.m;
#pragma unroll
              for (int j = 0; j < COB; ++j) acc[j] = fmaf(xv.v[c], Pk[j], acc[j]);
            }
          }
        } else {
          for (int k = 0; k < g.k; ++k) {
            const float xv = g.in_planar ? ld(x, ((size_t)n * g.k + k) * ivol + isp)
                                         : ld(x, ((size_t)n * ivol + isp) * g.k + k);
            const float* Pk = Pt + (size_t)k * g.m;
#pragma unroll
            for (int j = 0; j < COB; ++j) acc[j] = fmaf(xv, Pk[j], acc[j]);
          }
        }
      }
    }
  }
  if (g.out_planar) {
#pragma unroll
    for (int j = 0; j < COB; ++j) {
      const size_t o = ((size_t)n * g.m + m0 + j) * ovol + sp;
      st(y, o, acc[j] + (skip ? ld(skip, o) : 0.f));
    }
  } else {
    const size_t o = v * g.m + m0;
    if constexpr (COB % 8 == 0) {  // 16/32-byte vector stores: one instruction per 8 channels instead of 8
#pragma unroll
      for (int j0 = 0; j0 < COB; j0 += 8) {
        F8 ov;
        if (skip) {
          const F8 sk = ld8(skip, o + j0);
#pragma unroll
          for (int j = 0; j < 8; ++j) ov.v[j] = acc[j0 + j] + sk.v[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) ov.v[j] = acc[j0 + j];
        }
        st8(y, o + j0, ov);
      }
    } else {
#pragma unroll
      for (int j = 0; j < COB; ++j) st(y, o + j, acc[j] + (skip ? ld(skip, o + j) : 0.f));
    }
  }
}

template <typename TI, typename TO, int MAP>
static int launch_direct_cob(const void* x, const float* P, const float* bias, const void* skip, void* y,
                             const ConvGeom& g, hipStream_t s) {
  const size_t nvox = (size_t)g.n * g.od * g.oh * g.ow;
  const unsigned gx = (unsigned)((nvox + 255) / 256);
  const int m = g.m;
#define MEDNET_LAUNCH_COB(COB)                                                                                  \
  hipLaunchKernelGGL((conv_direct_kernel<TI, TO, MAP, COB>), dim3(gx, m / COB), dim3(256), 0, s, (const TI*)x, P, \
                     bias, (const TO*)skip, (TO*)y, g)
  if (m % 16 == 0) MEDNET_LAUNCH_COB(16);
  else if (m % 8 == 0) MEDNET_LAUNCH_COB(8);
  else if (m % 4 == 0) MEDNET_LAUNCH_COB(4);
  else if (m % 2 == 0) MEDNET_LAUNCH_COB(2);
  else MEDNET_LAUNCH_COB(1);
#undef MEDNET_LAUNCH_COB
  return check_launch("conv_direct");
}

template <int MAP>
int launch_direct(const void* x, const float* P, const float* bias, const void* skip, void* y, const ConvGeom& g,
                  int x_dtype, int y_dtype, hipStream_t s) {
  if (x_dtype == MEDNET_F32 && y_dtype == MEDNET_F32) return launch_direct_cob<float, float, MAP>(x, P, bias, skip, y, g, s);
  if (x_dtype == MEDNET_BF16 && y_dtype == MEDNET_BF16) return launch_direct_cob<bf16, bf16, MAP>(x, P, bias, skip, y, g, s);
  if (x_dtype == MEDNET_F32 && y_dtype == MEDNET_BF16) return launch_direct_cob<float, bf16, MAP>(x, P, bias, skip, y, g, s);
  if (x_dtype == MEDNET_BF16 && y_dtype == MEDNET_F32) return launch_direct_cob<bf16, float, MAP>(x, P, bias, skip, y, g, s);
  if (x_dtype == MEDNET_F16 && y_dtype == MEDNET_F16) return launch_direct_cob<f16, f16, MAP>(x, P, bias, skip, y, g, s);
  if (x_dtype == MEDNET_F32 && y_dtype == MEDNET_F16) return launch_direct_cob<float, f16, MAP>(x, P, bias, skip, y, g, s);
  if (x_dtype == MEDNET_F16 && y_dtype == MEDNET_F32) return launch_direct_cob<f16, float, MAP>(x, P, bias, skip, y, g, s);
  return fail(MEDNET_E_DTYPE, "conv_direct: dtype pair %d -> %d", x_dtype, y_dtype);
}
template int launch_direct<MAP_CONV>(const void*, const float*, const float*, const void*, void*, const ConvGeom&, int, int, hipStream_t);
template int launch_direct<MAP_CT_FWD>(const void*, const float*, const float*, const void*, void*, const ConvGeom&, int, int, hipStream_t);
template int launch_direct<MAP_CT_DG>(const void*, const float*, const float*, const void*, void*, const ConvGeom&, int, int, hipStream_t);

// ---- weight gradient: R[t][a][b] = sum_v A[v][a] * B[map(v,t)][b]  ->  dw[a][b][t] ---------------------------
//   conv  : A = dy (a = co), B = x  (b = ci), map = v + t - pad          dw = (Cout,Cin,k,k,k)
//   convT : A = x  (a = ci), B = dy (b = co), map = 2v - 1 + t           dw = (Cin,Cout,3,3,3)
// A workgroup owns a 16x16 (a,b) block and a chunk of voxels; each thread keeps the 27 taps of one (a,b) pair in
// registers.  Per-chunk partials go to the workspace and a second kernel sums them in a fixed order.

template <typename TA, typename TB, int KS>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(const TA* __restrict__ A, const TB* __restrict__ B,
                                                           float* __restrict__ part, WgradGeom g) {
  const int ta = threadIdx.x >> 4, tb = threadIdx.x & 15;
  const int nbb = (g.kb + 15) / 16;
  const int a = (blockIdx.y / nbb) * 16 + ta;
  const int b = (blockIdx.y % nbb) * 16 + tb;
  const bool live = a < g.ka && b < g.kb;
  constexpr int T = KS * KS * KS;
  const size_t avol = (size_t)g.ad * g.ah * g.aw, bvol = (size_t)g.bd * g.bh * g.bw;
  const size_t nvox = (size_t)g.n * avol;
  const size_t v0 = (size_t)blockIdx.x * g.chunk;
  const size_t v1 = v0 + g.chunk < nvox ? v0 + g.chunk : nvox;
  float acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.f;
  constexpr int pad = KS >> 1;
  if (live) {
    for (size_t v = v0; v < v1; ++v) {
      const int n = (int)(v / avol);
      const size_t sp = v - (size_t)n * avol;
      const int z = (int)(sp / ((size_t)g.ah * g.aw));
      const int rem = (int)(sp - (size_t)z * g.ah * g.aw);
      const int yy = rem / g.aw;
      const int xx = rem - yy * g.aw;
      const float av = g.a_planar ? ld(A, ((size_t)n * g.ka + a) * avol + sp) : ld(A, v * g.ka + a);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        {
          const int tz = t / (KS * KS), ty = (t / KS) % KS, tx = t % KS;
          const int bz = g.stride2 ? 2 * z - 1 + tz : z + tz - pad;
          const int by = g.stride2 ? 2 * yy - 1 + ty : yy + ty - pad;
          const int bx = g.stride2 ? 2 * xx - 1 + tx : xx + tx - pad;
          if (bz >= 0 && bz < g.bd && by >= 0 && by < g.bh && bx >= 0 && bx < g.bw) {
            const size_t bsp = ((size_t)bz * g.bh + by) * g.bw + bx;
            const float bv = g.b_planar ? ld(B, ((size_t)n * g.kb + b) * bvol + bsp)
                                        : ld(B, ((size_t)n * bvol + bsp) * g.kb + b);
            acc[t] = fmaf(av, bv, acc[t]);
          }
        }
      }
    }
    float* p = part + ((size_t)blockIdx.x * g.ka + a) * g.kb * T + (size_t)b * T;
#pragma unroll
    for (int t = 0; t < T; ++t) p[t] = acc[t];
  }
}

// dw[i] = sum_c part[c][i], fixed order (bitwise reproducible)
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            size_t count, int chunks) {
  // 64 outputs per workgroup; the 4 waves take every 4th chunk each, 4 loads in flight per thread (a single chain over
  // 1024 chunks was 85 us of pure latency for the 864 outputs of the first layer); fixed order => reproducible
  __shared__ float sh[4][64];
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < count) {
    int c = q;
    for (; c + 12 < chunks; c += 16) {
      s0 += part[(size_t)c * count + i];
      s1 += part[(size_t)(c + 4) * count + i];
      s2 += part[(size_t)(c + 8) * count + i];
      s3 += part[(size_t)(c + 12) * count + i];
    }
    for (; c < chunks; c += 4) s0 += part[(size_t)c * count + i];
  }
  sh[q][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && i < count) out[i] = (sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e]);
}

static void wgrad_plan(size_t nvox, int ka, int kb, size_t& chunk, unsigned& chunks) {
  const size_t blocks_ab = (size_t)((ka + 15) / 16) * ((kb + 15) / 16);
  size_t want = 4096 / blocks_ab;
  if (want < 1) want = 1;
  const size_t maxc = (nvox + 255) / 256;
  if (want > maxc) want = maxc;
  chunk = (nvox + want - 1) / want;
  chunks = (unsigned)((nvox + chunk - 1) / chunk);
}

size_t wgrad_direct_ws_bytes(size_t nvox, int ka, int kb, int ks) {
  size_t chunk;
  unsigned chunks;
  wgrad_plan(nvox, ka, kb, chunk, chunks);
  return (size_t)chunks * ka * kb * ks * ks * ks * sizeof(float);
}

int launch_wgrad_direct(const void* A, const void* B, float* dw, WgradGeom g, int a_dtype, int b_dtype, void* ws,
                        size_t ws_bytes, hipStream_t s) {
  const size_t nvox = (size_t)g.n * g.ad * g.ah * g.aw;
  unsigned chunks;
  wgrad_plan(nvox, g.ka, g.kb, g.chunk, chunks);
  const size_t count = (size_t)g.ka * g.kb * g.ks * g.ks * g.ks;
  MEDNET_REQUIRE(ws_bytes >= (size_t)chunks * count * sizeof(float), MEDNET_E_WORKSPACE,
                 "wgrad workspace too small: %zu < %zu", ws_bytes, (size_t)chunks * count * sizeof(float));
  const dim3 grid(chunks, ((g.ka + 15) / 16) * ((g.kb + 15) / 16));
  float* part = (float*)ws;
#define MEDNET_WG(TA_, TB_)                                                                                       \
  do {                                                                                                            \
    if (g.ks == 3)                                                                                                \
      hipLaunchKernelGGL((wgrad_direct_kernel<TA_, TB_, 3>), grid, dim3(256), 0, s, (const TA_*)A, (const TB_*)B, \
                         part, g);                                                                                \
    else                                                                                                          \
      hipLaunchKernelGGL((wgrad_direct_kernel<TA_, TB_, 1>), grid, dim3(256), 0, s, (const TA_*)A, (const TB_*)B, \
                         part, g);                                                                                \
  } while (0)
  if (a_dtype == MEDNET_F32 && b_dtype == MEDNET_F32) MEDNET_WG(float, float);
  else if (a_dtype == MEDNET_BF16 && b_dtype == MEDNET_BF16) MEDNET_WG(bf16, bf16);
  else if (a_dtype == MEDNET_F32 && b_dtype == MEDNET_BF16) MEDNET_WG(float, bf16);
  else if (a_dtype == MEDNET_BF16 && b_dtype == MEDNET_F32) MEDNET_WG(bf16, float);
  else if (a_dtype == MEDNET_F16 && b_dtype == MEDNET_F16) MEDNET_WG(f16, f16);
  else if (a_dtype == MEDNET_F32 && b_dtype == MEDNET_F16) MEDNET_WG(float, f16);
  else MEDNET_WG(f16, float);
#undef MEDNET_WG
  int rc = check_launch("wgrad_direct");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, s, part, dw, count,
                     (int)chunks);
  return check_launch("wgrad_reduce");
}

// ---- per-channel sum over voxels (bias gradients): per-workgroup partials, then a fixed-order combine -------------
constexpr int CS_ITEMS = 256 * 256;  // elements of one channel plane / of a channels-last slab per workgroup
// planar (NCDHW): grid (chunks, c, n); channels-last: grid (chunks, 1, n) with thread -> channel = tid % c
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const T* __restrict__ x, float* __restrict__ part,
                                                                  size_t spatial, int c, int planar) {
  __shared__ float scratch[256];
  const int n = blockIdx.z;
  if (planar) {
    const int ch = blockIdx.y;
    const T* p = x + ((size_t)n * c + ch) * spatial;
    const size_t v0 = (size_t)blockIdx.x * CS_ITEMS;
    const size_t v1 = v0 + CS_ITEMS < spatial ? v0 + CS_ITEMS : spatial;
    float s = 0.f;
    if (spatial % 8 == 0) {  // 8 elements (16 / 32 bytes) per load: one element per load ran at 0.5 TB/s
      float s1 = 0.f;
      for (size_t v = v0 + (size_t)threadIdx.x * 8; v + 8 <= v1; v += 256 * 8) {
        const F8 t = ld8(p, v);
        s += (t.v[0] + t.v[1]) + (t.v[2] + t.v[3]);
        s1 += (t.v[4] + t.v[5]) + (t.v[6] + t.v[7]);
      }
      s += s1;
    } else {
      for (size_t v = v0 + threadIdx.x; v < v1; v += 256) s += ld(p, v);
    }
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) part[((size_t)n * gridDim.x + blockIdx.x) * c + ch] = s;
  } else if (c % 8 == 0 && c <= 2048) {
    // a lane owns 8 consecutive channels (one 16-byte load per voxel); rows = 256 / (c/8) voxel rows per trip
    const int cols = c / 8, rows = 256 / cols, row = threadIdx.x / cols, col = threadIdx.x % cols;
    const bool active = row < rows;
    const size_t per = CS_ITEMS / c;
    const size_t b0 = (size_t)blockIdx.x * per;
    const size_t b1 = b0 + per < spatial ? b0 + per : spatial;
    float s8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s8[k] = 0.f;
    if (active)
      for (size_t v = b0 + row; v < b1; v += rows) {
        const F8 xv = ld8(x, ((size_t)n * spatial + v) * c + (size_t)col * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) s8[k] += xv.v[k];
      }
    __shared__ float red[256 * 8];
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = active ? s8[k] : 0.f;
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += 256) {
      float t = 0.f;
      for (int r = 0; r < rows; ++r) t += red[(r * cols + ch / 8) * 8 + ch % 8];
      part[((size_t)n * gridDim.x + blockIdx.x) * c + ch] = t;
    }
  } else if (c <= 256) {
    // thread t owns channel t % c; rows = 256 / c voxel rows are active
    const int rows = 256 / c, row = threadIdx.x / c, ch = threadIdx.x % c;
    const bool active = row < rows;
    const size_t per = CS_ITEMS / c;
    const size_t b0 = (size_t)blockIdx.x * per;
    const size_t b1 = b0 + per < spatial ? b0 + per : spatial;
    float s = 0.f;
    if (active)
      for (size_t v = b0 + row; v < b1; v += rows) s += ld(x, ((size_t)n * spatial + v) * c + ch);
    scratch[threadIdx.x] = active ? s : 0.f;
    __syncthreads();
    if ((int)threadIdx.x < c) {
      float t = 0.f;
      for (int r = 0; r < rows; ++r) t += scratch[r * c + threadIdx.x];
      part[((size_t)n * gridDim.x + blockIdx.x) * c + threadIdx.x] = t;
    }
  } else {
    // wide layers: a thread walks its channels (tid, tid + 256, ...) over the block's 32 voxels
    const size_t per = CS_ITEMS / 256;
    const size_t b0 = (size_t)blockIdx.x * per;
    const size_t b1 = b0 + per < spatial ? b0 + per : spatial;
    for (int ch = threadIdx.x; ch < c; ch += 256) {
      float s = 0.f;
      for (size_t v = b0; v < b1; ++v) s += ld(x, ((size_t)n * spatial + v) * c + ch);
      part[((size_t)n * gridDim.x + blockIdx.x) * c + ch] = s;
    }
  }
}
// one wave per channel: fp64 combine over (n, chunks)
__global__ __launch_bounds__(64) void channel_sum_final_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                               int items, int c) {
  const int ch = blockIdx.x;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four loads in flight (one dependent chain: 93 us for 4096 items)
  int i = threadIdx.x;
  for (; i + 192 < items; i += 256) {
    s0 += (double)part[(size_t)i * c + ch];
    s1 += (double)part[(size_t)(i + 64) * c + ch];
    s2 += (double)part[(size_t)(i + 128) * c + ch];
    s3 += (double)part[(size_t)(i + 192) * c + ch];
  }
  for (; i < items; i += 64) s0 += (double)part[(size_t)i * c + ch];
  const double s = wave_sum((s0 + s1) + (s2 + s3));
  if (threadIdx.x == 0) out[ch] = (float)s;
}

size_t channel_sum_ws_bytes(int n, size_t spatial, int c) {
  const size_t chunks_planar = (spatial + CS_ITEMS - 1) / CS_ITEMS;
  const size_t per = (c % 8 == 0 && c <= 2048) ? CS_ITEMS / c : CS_ITEMS / (c < 256 ? c : 256);
  const size_t chunks_cl = (spatial + per - 1) / per;
  const size_t chunks = chunks_planar > chunks_cl ? chunks_planar : chunks_cl;
  return (size_t)n * chunks * c * sizeof(float) + 256;
}

int launch_channel_sum(const void* x, float* out, int n, size_t spatial, int c, int planar, int dtype, void* ws,
                       size_t ws_bytes, hipStream_t s) {
  MEDNET_REQUIRE(ws_bytes >= channel_sum_ws_bytes(n, spatial, c), MEDNET_E_WORKSPACE, "channel_sum: workspace too small");
  const size_t per = planar ? CS_ITEMS : ((c % 8 == 0 && c <= 2048) ? CS_ITEMS / c : CS_ITEMS / (c < 256 ? c : 256));
  const unsigned chunks = (unsigned)((spatial + per - 1) / per);
  const dim3 grid(chunks, planar ? c : 1, n);
  float* part = (float*)ws;
  if (dtype == MEDNET_F32)
    hipLaunchKernelGGL(channel_sum_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)x, part, spatial, c, planar);
  else if (dtype == MEDNET_BF16)
    hipLaunchKernelGGL(channel_sum_partial_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x, part, spatial, c, planar);
  else
    hipLaunchKernelGGL(channel_sum_partial_kernel<f16>, grid, dim3(256), 0, s, (const f16*)x, part, spatial, c, planar);
  int rc = check_launch("channel_sum_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3(c), dim3(64), 0, s, part, out, (int)(n * chunks), c);
  return check_launch("channel_sum_final");
}

// ---- first-layer weight gradient (Cin == 1): dw[co][tap] = sum_v dy[v][co] * x[v + tap - 1] -------------------------
// The generic kernel would leave 15/16 of its threads idle here, and this layer sees the largest tensor of the net.
// A workgroup walks 4x8x32-voxel bricks: the brick's x halo (6x10x34 floats) sits in LDS, thread (co, g) keeps the 27
// taps of its output channel in registers and handles 4 consecutive voxels per step so that each LDS row read (6 floats)
// feeds 12 FMAs; dy is read straight from HBM, 32 consecutive channels per half-wave (64 B bf16 rows, coalesced).
constexpr int W1_TZ = 4, W1_TY = 8, W1_TX = 32;
constexpr int W1_HZ = W1_TZ + 2, W1_HY = W1_TY + 2, W1_HX = W1_TX + 2, W1_HXP = W1_HX + 2;  // row pitch 36 floats

template <typename TX_, typename TDY, int CO>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(const TX_* __restrict__ x, const TDY* __restrict__ dy,
                                                       float* __restrict__ part, int n, int d, int h, int w,
                                                       int tiles_z, int tiles_y, int tiles_x, int ntiles) {
  constexpr int G = 256 / CO;  // voxel groups per workgroup
  __shared__ float xs[W1_HZ * W1_HY * W1_HXP];
  __shared__ float red[256 * 27 / (CO >= 32 ? 1 : 1)];
  const int co = threadIdx.x % CO, g = threadIdx.x / CO;
  float acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = 0.f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int tt = tile;
    const int x0 = (tt % tiles_x) * W1_TX;
    tt /= tiles_x;
    const int y0 = (tt % tiles_y) * W1_TY;
    tt /= tiles_y;
    const int z0 = (tt % tiles_z) * W1_TZ;
    const int nn = tt / tiles_z;
    __syncthreads();
    for (int i = threadIdx.x; i < W1_HZ * W1_HY * W1_HX; i += 256) {
      const int hx = i % W1_HX, hy = (i / W1_HX) % W1_HY, hz = i / (W1_HX * W1_HY);
      const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      float v = 0.f;
      if (gz >= 0 && gz < d && gy >= 0 && gy < h && gx >= 0 && gx < w) v = ld(x, (((size_t)nn * d + gz) * h + gy) * w + gx);
      xs[(hz * W1_HY + hy) * W1_HXP + hx] = v;
    }
    __syncthreads();
    // quads of 4 consecutive x voxels: 4*8*8 = 256 quads per brick
    for (int qd = g; qd < W1_TZ * W1_TY * (W1_TX / 4); qd += G) {
      const int qx = (qd % (W1_TX / 4)) * 4, qy = (qd / (W1_TX / 4)) % W1_TY, qz = qd / ((W1_TX / 4) * W1_TY);
      const int gz = z0 + qz, gy = y0 + qy;
      float dv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int gx = x0 + qx + k;
        dv[k] = (gz < d && gy < h && gx < w) ? ld(dy, ((((size_t)nn * d + gz) * h + gy) * w + gx) * CO + co) : 0.f;
      }
#pragma unroll
      for (int tz = 0; tz < 3; ++tz)
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const float* row = xs + ((qz + tz) * W1_HY + qy + ty) * W1_HXP + qx;
          float r[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) r[k] = row[k];
#pragma unroll
          for (int tx = 0; tx < 3; ++tx)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[(tz * 3 + ty) * 3 + tx] = fmaf(dv[k], r[k + tx], acc[(tz * 3 + ty) * 3 + tx]);
        }
    }
  }
  // sum the G voxel groups of each channel, then one partial per workgroup
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 27; ++t) red[(t * G + g) * CO + co] = acc[t];
  __syncthreads();
  for (int i = threadIdx.x; i < 27 * CO; i += 256) {
    const int t = i / CO, c2 = i % CO;
    float s = 0.f;
    for (int k = 0; k < G; ++k) s += red[(t * G + k) * CO + c2];
    part[(size_t)blockIdx.x * (27 * CO) + (size_t)c2 * 27 + t] = s;
  }
}

// ---- 1x1x1 head data gradient: dz[v][k] = sum_m dy[m][v] * W[m][k]  (planar fp32 logits gradient -> channels-last) -----
// HBM-bound (writes the full-resolution 32-channel tensor once): a lane owns 8 consecutive channels of one voxel, so a
// wave stores 1 KB contiguous per instruction; the M x 8 weights of the lane's column live in registers.
template <typename TO, int MMAX>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ Pb /*[m][k]*/,
                                                         TO* __restrict__ dz, size_t spatial, int k, int m, size_t chunk_vox) {
  const int cols = k / 8, rows = 256 / cols;
  const int col = threadIdx.x % cols, row = threadIdx.x / cols;
  if ((int)threadIdx.x >= rows * cols) return;
  const int n = blockIdx.y;
  float w[MMAX][8];
#pragma unroll
  for (int i = 0; i < MMAX; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) w[i][j] = i < m ? Pb[(size_t)i * k + col * 8 + j] : 0.f;
  const size_t v0 = (size_t)blockIdx.x * chunk_vox;
  const size_t v1 = v0 + chunk_vox < spatial ? v0 + chunk_vox : spatial;
  for (size_t v = v0 + row; v < v1; v += rows) {
    F8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = 0.f;
#pragma unroll
    for (int i = 0; i < MMAX; ++i) {
      if (i < m) {
        const float d = dy[((size_t)n * m + i) * spatial + v];
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = fmaf(d, w[i][j], o.v[j]);
      }
    }
    st8(dz, ((size_t)n * spatial + v) * k + col * 8, o);
  }
}
// ---- 1x1x1 head, one voxel per lane (any class count; K = 16/32/64 feature channels) -------------------------------
// The weights are wave-uniform, so every FMA takes its weight from an SGPR (scalar loads, no LDS, no per-lane weight
// registers -- the column-register kernel above spills once M x 8 weights no longer fit, e.g. the 18 outputs of the
// landmark net).  Planar fp32 logits (or their gradient) are read / written fully coalesced: consecutive lanes are
// consecutive voxels of one class plane.  Both kernels are HBM-bound (one pass over the K-channel tensor and the planes).
template <typename TI, int K>
__global__ __launch_bounds__(256) void head_fwd_vox_kernel(const TI* __restrict__ z, const float* __restrict__ Pb /*[m][K]*/,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           size_t spatial, int m) {
  const int n = blockIdx.y;
  const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= spatial) return;
  float zv[K];
#pragma unroll
  for (int q = 0; q < K / 8; ++q) {
    const F8 t = ld8(z, ((size_t)n * spatial + v) * K + q * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) zv[q * 8 + j] = t.v[j];
  }
  for (int i = 0; i < m; ++i) {
    float s0 = bias ? bias[i] : 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < K; j += 2) {
      s0 = fmaf(zv[j], Pb[(size_t)i * K + j], s0);
      s1 = fmaf(zv[j + 1], Pb[(size_t)i * K + j + 1], s1);
    }
    y[((size_t)n * m + i) * spatial + v] = s0 + s1;
  }
}
template <typename TO, int K>
__global__ __launch_bounds__(256) void head_dgrad_vox_kernel(const float* __restrict__ dy, const float* __restrict__ Pb /*[m][K]*/,
                                                             TO* __restrict__ dz, size_t spatial, int m) {
  const int n = blockIdx.y;
  const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= spatial) return;
  float o[K];
#pragma unroll
  for (int j = 0; j < K; ++j) o[j] = 0.f;
#pragma unroll 4
  for (int i = 0; i < m; ++i) {
    const float d = dy[((size_t)n * m + i) * spatial + v];
#pragma unroll
    for (int j = 0; j < K; ++j) o[j] = fmaf(d, Pb[(size_t)i * K + j], o[j]);
  }
#pragma unroll
  for (int q = 0; q < K / 8; ++q) {
    F8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t.v[j] = o[q * 8 + j];
    st8(dz, ((size_t)n * spatial + v) * K + q * 8, t);
  }
}
// The same data gradient when its output dz is the gradient of an ExtResNetBlock's output (the last decoder block feeds
// the head, model.py:204-207): the block's backward starts with GroupNorm-3's first pass over (dz, out, y3)
// (components.py:170-178) -- du = dz * act'(out), sums of du and du * y3 per channel -- which is taken here from the STORED
// (rounded) dz instead of a stand-alone pass that reads dz again: partial[n][4 * workgroup + wave][K][2] =
// {sum du, sum du * y3}.
// Lane mapping: K/8 lanes share a voxel, each owns 8 channels (one 16-byte piece of the voxel's row), so a wave's loads and
// stores are whole contiguous rows (thread = voxel made every load touch 64 different cache lines: 534 us for 1.75 GB);
// a lane keeps 16 sums, lanes of one channel group are summed with DPP (LDS-free) and every wave writes one partial row.
constexpr int HEAD_GN_VPT = 32;  // voxels per lane group
constexpr int HEAD_GN_MAXM = 64;  // classes (LDS image of the weights)
constexpr int HEAD_DPP_MAXM = 32;  // classes the lanes of a voxel split between them (registers: HEAD_DPP_MAXM / lanes per voxel)
// MANY: the form for more than 8 classes, its own instantiation (the 4-class head of the segmentation nets keeps its registers)
template <typename TO, int K, int MR>  // MR: classes whose weights a lane keeps in registers (4 or 8), 0 = the form for more
__global__ __launch_bounds__(256) void head_dgrad_gn_kernel(const float* __restrict__ dy, const float* __restrict__ Pb /*[m][K]*/,
                                                            TO* __restrict__ dz, const TO* __restrict__ gy,
                                                            const TO* __restrict__ gz, float* __restrict__ partial,
                                                            size_t spatial, int m, int act) {
  constexpr bool MANY = MR == 0;
  constexpr int CG = K / 8, VPW = 256 / CG;  // lanes per voxel, voxels per workgroup pass
  const int n = blockIdx.y;
  const int cgi = threadIdx.x % CG, vi = threadIdx.x / CG;
  // the lane's slice of the weights (8 channels x m classes): in registers for up to 8 classes, else through LDS (per-lane
  // global loads inside the class loop made the 18-class landmark head compute-bound: 986 us)
  __shared__ float sPb[HEAD_GN_MAXM * K];
  float wreg[MANY ? 1 : MR][8];
  constexpr bool in_regs = !MANY;
  if constexpr (in_regs) {
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) wreg[i][j] = i < m ? Pb[(size_t)i * K + cgi * 8 + j] : 0.f;
  } else if (m <= HEAD_GN_MAXM) {
    for (int e = threadIdx.x; e < m * K; e += 256) sPb[e] = Pb[e];
    __syncthreads();
  }
  const float* wsrc = m <= HEAD_GN_MAXM ? sPb : Pb;  // (more classes than the LDS image holds: straight from memory)
  float ss[8], sq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ss[j] = sq[j] = 0.f;
  const size_t v0 = (size_t)blockIdx.x * (VPW * HEAD_GN_VPT) + vi;
  for (int it = 0; it < HEAD_GN_VPT; ++it) {
    const size_t v = v0 + (size_t)it * VPW;
    if (v >= spatial) break;
    F8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t.v[j] = 0.f;
    if constexpr (in_regs) {
      // TWO voxels per trip, every load of both (logit gradients, block output, GroupNorm input) issued before the first use:
      // one voxel per trip at 4 waves per SIMD kept too little in flight (581 us for 1.75 GB in the bf16 step: 3.0 TB/s)
      const size_t vb = v + VPW;
      const bool hb = (it + 1 < HEAD_GN_VPT) && vb < spatial;
      float da[MANY ? 1 : MR], db[MANY ? 1 : MR];
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        da[i] = i < m ? dy[((size_t)n * m + i) * spatial + v] : 0.f;
        db[i] = (i < m && hb) ? dy[((size_t)n * m + i) * spatial + vb] : 0.f;
      }
      const size_t rowa = ((size_t)n * spatial + v) * K + cgi * 8, rowb = ((size_t)n * spatial + (hb ? vb : v)) * K + cgi * 8;
      const F8 zva = ld8(gz, rowa), yva = ld8(gy, rowa), zvb = ld8(gz, rowb), yvb = ld8(gy, rowb);
      F8 u;
#pragma unroll
      for (int j = 0; j < 8; ++j) u.v[j] = 0.f;
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        if (i < m) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            t.v[j] = fmaf(da[i], wreg[i][j], t.v[j]);
            u.v[j] = fmaf(db[i], wreg[i][j], u.v[j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {  // the stored value is what GroupNorm-3's second pass reads
        t.v[j] = (float)(TO)t.v[j];
        u.v[j] = (float)(TO)u.v[j];
      }
      st8(dz, rowa, t);
      if (hb) st8(dz, rowb, u);
      act_grad_n<8>(t.v, zva.v, act);
      act_grad_n<8>(u.v, zvb.v, act);  // (no second voxel: u = 0, and act' of every kind keeps a zero gradient zero)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ss[j] += t.v[j] + u.v[j];
        sq[j] = fmaf(u.v[j], yvb.v[j], fmaf(t.v[j], yva.v[j], sq[j]));
      }
      ++it;  // (the trip took two voxels)
      continue;
    } else if (CG <= 4 && m <= HEAD_DPP_MAXM) {
      // More than 8 classes (the landmark head: 16 heat maps + 2 classes, landmarks.py:71-75).  The CG lanes of a voxel split
      // the classes between them -- lane c loads classes c, c + CG, ... -- and hand the values round with DPP quad
      // permutes: ceil(m / CG) planar loads per lane, ALL in flight before the first use, instead of m loads of which two
      // were in flight (18 classes: 986 us; the loop was latency-bound, not bandwidth-bound).
      // TWO voxels per trip: each weight row comes out of LDS once for both (the 36 ds_read_b128 per voxel were a third of the
      // kernel's time), and the rows of the block output / the GroupNorm input of both voxels are in flight with the logit
      // gradients.
      constexpr int MAXQ = HEAD_DPP_MAXM / CG;
      const size_t vb = v + VPW;
      const bool hb = (it + 1 < HEAD_GN_VPT) && vb < spatial;
      float dreg[MAXQ], dregb[MAXQ];
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) {
        const int i = q * CG + cgi;
        const bool live = q * CG < m && i < m;
        dreg[q] = live ? dy[((size_t)n * m + i) * spatial + v] : 0.f;
        dregb[q] = (live && hb) ? dy[((size_t)n * m + i) * spatial + vb] : 0.f;
      }
      const size_t rowa = ((size_t)n * spatial + v) * K + cgi * 8, rowb = ((size_t)n * spatial + (hb ? vb : v)) * K + cgi * 8;
      const F8 zva = ld8(gz, rowa), yva = ld8(gy, rowa), zvb = ld8(gz, rowb), yvb = ld8(gy, rowb);
      F8 u;
#pragma unroll
      for (int j = 0; j < 8; ++j) u.v[j] = 0.f;
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) {
        if (q * CG < m) {  // (wave-uniform)
#pragma unroll
          for (int c = 0; c < CG; ++c) {
            const int i = q * CG + c;
            if (i < m) {
              // broadcast lane c of every CG-lane group: quad_perm [c,c,c,c] (CG = 4) or [c,c,2+c,2+c] (CG = 2)
              constexpr int HI = CG == 4 ? 0 : 2;
              auto bc = [&](float x) {
                return c == 0 ? dpp_f32<(0) | (0 << 2) | ((HI + 0) << 4) | ((HI + 0) << 6)>(x)
                     : c == 1 ? dpp_f32<(1) | (1 << 2) | ((HI + 1) << 4) | ((HI + 1) << 6)>(x)
                     : c == 2 ? dpp_f32<(2) | (2 << 2) | (2 << 4) | (2 << 6)>(x)
                              : dpp_f32<(3) | (3 << 2) | (3 << 4) | (3 << 6)>(x);
              };
              const float d = bc(dreg[q]), db = bc(dregb[q]);
              const f32x4 w0 = *reinterpret_cast<const f32x4*>(wsrc + i * K + cgi * 8), w1 = *reinterpret_cast<const f32x4*>(wsrc + i * K + cgi * 8 + 4);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                t.v[j] = fmaf(d, w0[j], t.v[j]);
                t.v[4 + j] = fmaf(d, w1[j], t.v[4 + j]);
                u.v[j] = fmaf(db, w0[j], u.v[j]);
                u.v[4 + j] = fmaf(db, w1[j], u.v[4 + j]);
              }
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {  // the stored value is what GroupNorm-3's second pass reads
        t.v[j] = (float)(TO)t.v[j];
        u.v[j] = (float)(TO)u.v[j];
      }
      st8(dz, rowa, t);
      if (hb) st8(dz, rowb, u);
      act_grad_n<8>(t.v, zva.v, act);
      act_grad_n<8>(u.v, zvb.v, act);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ss[j] += t.v[j] + u.v[j];
        sq[j] = fmaf(u.v[j], yvb.v[j], fmaf(t.v[j], yva.v[j], sq[j]));
      }
      ++it;  // (the trip took two voxels)
      continue;
    } else {
#pragma unroll 2
      for (int i = 0; i < m; ++i) {
        const float d = dy[((size_t)n * m + i) * spatial + v];
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wsrc + i * K + cgi * 8), w1 = *reinterpret_cast<const f32x4*>(wsrc + i * K + cgi * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t.v[j] = fmaf(d, w0[j], t.v[j]);
          t.v[4 + j] = fmaf(d, w1[j], t.v[4 + j]);
        }
      }
    }
    const size_t row = ((size_t)n * spatial + v) * K + cgi * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) t.v[j] = (float)(TO)t.v[j];  // the stored value is what GroupNorm-3's second pass reads
    st8(dz, row, t);
    const F8 zv = ld8(gz, row), yv = ld8(gy, row);
    act_grad_n<8>(t.v, zv.v, act);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ss[j] += t.v[j];
      sq[j] = fmaf(t.v[j], yv.v[j], sq[j]);
    }
  }
  float* out = partial + (((size_t)n * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * K * 2 + cgi * 16;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = lane_class_sum<CG>(ss[j]), b = lane_class_sum<CG>(sq[j]);
    if ((threadIdx.x & 63) < CG) {
      out[2 * j] = a;
      out[2 * j + 1] = b;
    }
  }
}
int head_dgrad_gn_rows(size_t spatial, int k, int dtype) {
  if (!(k == 16 || k == 32 || k == 64) || !dtype_ok(dtype)) return 0;  // (fp32 storage too, round 3)
  const size_t per_wg = (size_t)(256 / (k / 8)) * HEAD_GN_VPT;
  return 4 * (int)((spatial + per_wg - 1) / per_wg);
}
int launch_head_dgrad_gn(const void* dy, const float* Pb, void* dz, const void* gy, const void* gz, int act, float* partial,
                         int n, size_t spatial, int m, int k, int dtype, hipStream_t s) {
  MEDNET_REQUIRE(head_dgrad_gn_rows(spatial, k, dtype) > 0, MEDNET_E_UNSUPPORTED, "head_dgrad_gn: K=%d dtype=%d", k, dtype);
  const dim3 grid((unsigned)(head_dgrad_gn_rows(spatial, k, dtype) / 4), n);
#define HG_GO(TO_, K_)                                                                                                         \
  do {                                                                                                                         \
    if (m <= 4)                                                                                                                \
      hipLaunchKernelGGL((head_dgrad_gn_kernel<TO_, K_, 4>), grid, dim3(256), 0, s, (const float*)dy, Pb, (TO_*)dz,            \
                         (const TO_*)gy, (const TO_*)gz, partial, spatial, m, act);                                            \
    else if (m <= 8)                                                                                                           \
      hipLaunchKernelGGL((head_dgrad_gn_kernel<TO_, K_, 8>), grid, dim3(256), 0, s, (const float*)dy, Pb, (TO_*)dz,            \
                         (const TO_*)gy, (const TO_*)gz, partial, spatial, m, act);                                            \
    else                                                                                                                       \
      hipLaunchKernelGGL((head_dgrad_gn_kernel<TO_, K_, 0>), grid, dim3(256), 0, s, (const float*)dy, Pb, (TO_*)dz,            \
                         (const TO_*)gy, (const TO_*)gz, partial, spatial, m, act);                                            \
  } while (0)
#define HG_K(TO_) do { if (k == 16) HG_GO(TO_, 16); else if (k == 32) HG_GO(TO_, 32); else HG_GO(TO_, 64); } while (0)
  if (dtype == MEDNET_F32) HG_K(float);
  else if (dtype == MEDNET_BF16) HG_K(bf16);
  else HG_K(f16);
#undef HG_K
#undef HG_GO
  return check_launch("head_dgrad_gn");
}
bool head_vox_supported(int k) { return k == 16 || k == 32 || k == 64; }
int launch_head_fwd_vox(const void* z, const float* Pb, const float* bias, float* y, int n, size_t spatial, int k, int m,
                        int z_dtype, hipStream_t s) {
  const dim3 grid((unsigned)((spatial + 255) / 256), n);
#define HF_GO(TI_, K_) hipLaunchKernelGGL((head_fwd_vox_kernel<TI_, K_>), grid, dim3(256), 0, s, (const TI_*)z, Pb, bias, y, spatial, m)
#define HF_K(TI_) do { if (k == 16) HF_GO(TI_, 16); else if (k == 32) HF_GO(TI_, 32); else HF_GO(TI_, 64); } while (0)
  if (z_dtype == MEDNET_F32) HF_K(float);
  else if (z_dtype == MEDNET_BF16) HF_K(bf16);
  else HF_K(f16);
#undef HF_K
#undef HF_GO
  return check_launch("head_fwd_vox");
}

bool head_dgrad_supported(int cin /*dy channels = layer Cout*/, int cout /*dz channels = layer Cin*/, int ksize, int x_dtype,
                          int x_layout, int y_layout) {
  return ksize == 1 && x_dtype == MEDNET_F32 && x_layout == MEDNET_NCDHW && y_layout == MEDNET_NDHWC && cout % 8 == 0 &&
         cout / 8 <= 256 && (cin <= 32 || head_vox_supported(cout));
}
int launch_head_dgrad(const void* dy, const float* Pb, void* dz, int n, size_t spatial, int m, int k, int out_dtype,
                      hipStream_t s) {
  if (head_vox_supported(k)) {
    const dim3 grid((unsigned)((spatial + 255) / 256), n);
#define HV_GO(TO_, K_) hipLaunchKernelGGL((head_dgrad_vox_kernel<TO_, K_>), grid, dim3(256), 0, s, (const float*)dy, Pb, (TO_*)dz, spatial, m)
#define HV_K(TO_) do { if (k == 16) HV_GO(TO_, 16); else if (k == 32) HV_GO(TO_, 32); else HV_GO(TO_, 64); } while (0)
    if (out_dtype == MEDNET_F32) HV_K(float);
    else if (out_dtype == MEDNET_BF16) HV_K(bf16);
    else HV_K(f16);
#undef HV_K
#undef HV_GO
    return check_launch("head_dgrad_vox");
  }
  const int rows = 256 / (k / 8);
  size_t cv = (spatial + 1023) / 1024;
  if (cv < (size_t)rows * 8) cv = (size_t)rows * 8;
  cv = (cv + rows - 1) / rows * rows;
  const dim3 grid((unsigned)((spatial + cv - 1) / cv), n);
#define HD_GO(TO_, MM_) hipLaunchKernelGGL((head_dgrad_kernel<TO_, MM_>), grid, dim3(256), 0, s, (const float*)dy, Pb, (TO_*)dz, spatial, k, m, cv)
#define HD_M(TO_)                     \
  do {                                \
    if (m <= 4) HD_GO(TO_, 4);        \
    else if (m <= 8) HD_GO(TO_, 8);   \
    else if (m <= 16) HD_GO(TO_, 16); \
    else HD_GO(TO_, 32);              \
  } while (0)
  if (out_dtype == MEDNET_F32) HD_M(float);
  else if (out_dtype == MEDNET_BF16) HD_M(bf16);
  else HD_M(f16);
#undef HD_M
#undef HD_GO
  return check_launch("head_dgrad");
}

// ---- 1x1x1 head weight gradient: dw[m][k] = sum_v dy[m][v] (planar fp32 logits gradient) * z[v][k] (channels-last) ----
// HBM-bound (one pass over z): a thread owns 8 input channels of a voxel column ("column persistent", like GroupNorm) and
// an 8-row block of output channels; per-workgroup partials, fixed-order combine.
// MB: classes of a block that are alive (4: the segmentation heads -- half the sums and four voxels per trip instead of two;
// 8 otherwise).  The partial rows keep the 8-per-block layout either way.
template <typename TZ, int MB>
__global__ __launch_bounds__(256) void wgrad_1x1_kernel(const float* __restrict__ dy, const TZ* __restrict__ z,
                                                        float* __restrict__ part, size_t spatial, int k, int m,
                                                        size_t chunk_vox) {
  __shared__ float lds[256 * 8];
  const int cols = k / 8, rows = 256 / cols;
  const int col = threadIdx.x % cols, row = threadIdx.x / cols;
  const bool active = (int)threadIdx.x < rows * cols;
  // grid = (class blocks, chunks, n): the workgroups that read the SAME chunk of z for different class blocks (the landmark
  // head has three) are dispatched next to each other, so z comes from HBM once and from the Infinity Cache after that
  const int n = blockIdx.z, m0 = blockIdx.x * 8;
  const size_t v0 = (size_t)blockIdx.y * chunk_vox;
  const size_t v1 = v0 + chunk_vox < spatial ? v0 + chunk_vox : spatial;
  float acc[MB][8];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  if (active) {
    const TZ* zb = z + (size_t)n * spatial * k + (size_t)col * 8;
    // VT voxels per trip, all their loads issued before the first use (one voxel per trip left the pass latency-bound:
    // 2.4 TB/s)
    constexpr int VT = 16 / MB;
    const float* dyb = dy + ((size_t)n * m + m0) * spatial;
    size_t v = v0 + row;
    for (; v + (size_t)(VT - 1) * rows < v1; v += (size_t)VT * rows) {
      F8 zv[VT];
      float d[VT][MB];
#pragma unroll
      for (int t = 0; t < VT; ++t) zv[t] = ld8(zb, (v + (size_t)t * rows) * k);
      if (cols == 4) {
        // The four lanes of a voxel split the classes between them (lane c loads classes c, c + 4) and hand the values round
        // with DPP quad permutes: MB / 4 planar 4-byte loads per lane and voxel instead of MB (every one of those is a whole
        // wave instruction for 64 useful bytes).  By itself this did not move the 18-class landmark head (758 us with three
        // class blocks); the column sums at the end of the kernel did (492 us), see below.
        float own[VT][MB / 4];
#pragma unroll
        for (int t = 0; t < VT; ++t)
#pragma unroll
          for (int q = 0; q < MB / 4; ++q) {
            const int i = q * 4 + col;
            own[t][q] = (m0 + i < m) ? dyb[(size_t)i * spatial + v + (size_t)t * rows] : 0.f;
          }
#pragma unroll
        for (int t = 0; t < VT; ++t)
#pragma unroll
          for (int q = 0; q < MB / 4; ++q) {
            d[t][q * 4 + 0] = dpp_f32<(0) | (0 << 2) | (0 << 4) | (0 << 6)>(own[t][q]);
            d[t][q * 4 + 1] = dpp_f32<(1) | (1 << 2) | (1 << 4) | (1 << 6)>(own[t][q]);
            d[t][q * 4 + 2] = dpp_f32<(2) | (2 << 2) | (2 << 4) | (2 << 6)>(own[t][q]);
            d[t][q * 4 + 3] = dpp_f32<(3) | (3 << 2) | (3 << 4) | (3 << 6)>(own[t][q]);
          }
      } else {
#pragma unroll
        for (int t = 0; t < VT; ++t)
#pragma unroll
          for (int i = 0; i < MB; ++i) d[t][i] = (m0 + i < m) ? dyb[(size_t)i * spatial + v + (size_t)t * rows] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < VT; ++t)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(d[t][i], zv[t].v[j], acc[i][j]);
    }
    for (; v < v1; v += rows) {
      const F8 zv = ld8(zb, v * k);
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const float d = (m0 + i < m) ? dyb[(size_t)i * spatial + v] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(d, zv.v[j], acc[i][j]);
      }
    }
  }
  float* out = part + (((size_t)n * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 * k;
  if (cols <= 8 && (cols & (cols - 1)) == 0) {
    // The threads that share a column: first the 64 / cols of a wave with shuffles (fixed tree), then the four waves through LDS,
    // every output summed by a thread of its own.  (One output row at a time with the column's `rows` values added up by ONE
    // thread -- the form below -- was 512 dependent LDS reads per class: with 24 workgroups per CU it cost the 18-class head
    // more than its pass over the data.)
    for (int off = cols; off < 64; off <<= 1) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] += __shfl_xor(acc[i][j], off, 64);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane < cols) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) lds[((wv * cols + lane) * MB + i) * 8 + j] = acc[i][j];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < cols * MB * 8; o += 256) {  // o = (column, class, channel of the column)
      const float sum = (lds[o] + lds[cols * MB * 8 + o]) + (lds[2 * cols * MB * 8 + o] + lds[3 * cols * MB * 8 + o]);
      const int j = o % 8, i = (o / 8) % MB, cc = o / (8 * MB);
      out[(size_t)i * k + cc * 8 + j] = sum;
    }
    return;
  }
  // (other column counts) reduce the `rows` threads that share a column, one output row at a time
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < 8; ++j) lds[(row * cols + col) * 8 + j] = acc[i][j];
    }
    __syncthreads();
    if (active && row == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += lds[(r * cols + col) * 8 + j];
        out[(size_t)i * k + col * 8 + j] = s;
      }
    }
  }
}
// dw[mm][kk] = sum over (n, chunk) of part[n][chunk][mb][i][kk]
__global__ __launch_bounds__(64) void wgrad_1x1_final_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                             int items, int mblocks, int k, int m) {
  const int mm = blockIdx.x / k, kk = blockIdx.x % k;
  const int mb = mm / 8, i = mm % 8;
  double s = 0.0;
  for (int it = threadIdx.x; it < items; it += 64) s += (double)part[(((size_t)it * mblocks + mb) * 8 + i) * k + kk];
  s = wave_sum(s);
  if (threadIdx.x == 0) dw[(size_t)mm * k + kk] = (float)s;
}
bool wgrad_1x1_supported(int cin, int cout, int ksize, int x_layout, int dy_layout, int dy_dtype) {
  return ksize == 1 && cin % 8 == 0 && cin / 8 <= 256 && x_layout == MEDNET_NDHWC && dy_layout == MEDNET_NCDHW &&
         dy_dtype == MEDNET_F32 && cout <= 4096;
}
static void wgrad_1x1_plan(size_t spatial, int k, size_t& chunk_vox, unsigned& chunks) {
  const int rows = 256 / (k / 8);
  size_t cv = (spatial + 511) / 512;
  if (cv < (size_t)rows * 8) cv = (size_t)rows * 8;
  cv = (cv + rows - 1) / rows * rows;
  chunk_vox = cv;
  chunks = (unsigned)((spatial + cv - 1) / cv);
}
size_t wgrad_1x1_ws_bytes(int n, size_t spatial, int cin, int cout) {
  if (cin % 8 || cin / 8 > 256) return 0;
  size_t cv;
  unsigned chunks;
  wgrad_1x1_plan(spatial, cin, cv, chunks);
  return (size_t)n * chunks * ((cout + 7) / 8) * 8 * cin * sizeof(float);
}
int launch_wgrad_1x1(const void* z, const void* dy, float* dw, int n, size_t spatial, int cin, int cout, int z_dtype,
                     void* ws, size_t ws_bytes, hipStream_t s) {
  size_t cv;
  unsigned chunks;
  wgrad_1x1_plan(spatial, cin, cv, chunks);
  const int mblocks = (cout + 7) / 8;
  MEDNET_REQUIRE(ws_bytes >= wgrad_1x1_ws_bytes(n, spatial, cin, cout), MEDNET_E_WORKSPACE, "wgrad_1x1: workspace too small");
  const dim3 grid(mblocks, chunks, n);
  float* part = (float*)ws;
#define W11_GO(TZ_)                                                                                                                   \
  do {                                                                                                                                \
    if (cout <= 4)                                                                                                                    \
      hipLaunchKernelGGL((wgrad_1x1_kernel<TZ_, 4>), grid, dim3(256), 0, s, (const float*)dy, (const TZ_*)z, part, spatial, cin, cout, cv); \
    else                                                                                                                              \
      hipLaunchKernelGGL((wgrad_1x1_kernel<TZ_, 8>), grid, dim3(256), 0, s, (const float*)dy, (const TZ_*)z, part, spatial, cin, cout, cv); \
  } while (0)
  if (z_dtype == MEDNET_F32) W11_GO(float);
  else if (z_dtype == MEDNET_BF16) W11_GO(bf16);
  else W11_GO(f16);
#undef W11_GO
  int rc = check_launch("wgrad_1x1");
  if (rc) return rc;
  hipLaunchKernelGGL(wgrad_1x1_final_kernel, dim3(cout * cin), dim3(64), 0, s, part, dw, (int)(n * chunks), mblocks, cin, cout);
  return check_launch("wgrad_1x1_final");
}

bool wgrad_c1_supported(int cin, int cout, int ksize, int x_layout, int dy_layout) {
  return cin == 1 && ksize == 3 && (cout == 8 || cout == 16 || cout == 32 || cout == 64) && dy_layout == MEDNET_NDHWC &&
         (x_layout == MEDNET_NDHWC || x_layout == MEDNET_NCDHW);
}
static int wgrad_c1_blocks(int ntiles) { return ntiles < 1024 ? ntiles : 1024; }
size_t wgrad_c1_ws_bytes(int n, int d, int h, int w, int cout) {
  const int nt = n * ((d + W1_TZ - 1) / W1_TZ) * ((h + W1_TY - 1) / W1_TY) * ((w + W1_TX - 1) / W1_TX);
  const int b1 = wgrad_c1_blocks(nt), b2 = wgrad_c1_mfma_blocks(n, d, h, w);  // VALU and matrix-core forms
  return (size_t)(b1 > b2 ? b1 : b2) * 27 * cout * sizeof(float);
}
// The first layer's weight gradient with GroupNorm's backward applied while dz is staged (16-bit storage, matrix-core
// kernel only): dy is never materialised.  See Wc1Args in conv_mfma.hip.
bool wgrad_c1_gn_supported(int cout, int x_dtype, int dtype) {
  return tuning_option("wgrad_c1_mfma", 1) && (dtype == MEDNET_F16 ? mednet_f16::wgrad_c1_mfma_supported(cout, x_dtype, dtype)
                                                                  : wgrad_c1_mfma_supported(cout, x_dtype, dtype));
}
int launch_wgrad_c1_gn(const void* x, const void* dz, const void* y, const float* coef, const float* bcoef, int act, float* dw,
                       int n, int d, int h, int w, int cout, int x_dtype, int dtype, void* ws, size_t ws_bytes, hipStream_t s) {
  MEDNET_REQUIRE(wgrad_c1_gn_supported(cout, x_dtype, dtype), MEDNET_E_UNSUPPORTED,
                 "wgrad_c1_gn: Cout in {16, 32, 64}, 16-bit dz / y, x fp32 or the same 16-bit type (Cout=%d)", cout);
  const int blocks = wgrad_c1_mfma_blocks(n, d, h, w);
  MEDNET_REQUIRE(ws_bytes >= (size_t)blocks * 27 * cout * sizeof(float), MEDNET_E_WORKSPACE, "wgrad_c1_gn: workspace too small");
  float* part = (float*)ws;
  const int rc = dtype == MEDNET_F16 ? mednet_f16::launch_wgrad_c1_mfma(x, dz, part, n, d, h, w, cout, s, x_dtype, y, coef, bcoef, act)
                                     : launch_wgrad_c1_mfma(x, dz, part, n, d, h, w, cout, s, x_dtype, y, coef, bcoef, act);
  if (rc) return rc;
  const size_t count = (size_t)27 * cout;
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, s, part, dw, count, blocks);
  return check_launch("wgrad_c1_reduce");
}
int launch_wgrad_c1(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cout, int x_dtype,
                    int dy_dtype, void* ws, size_t ws_bytes, hipStream_t s, bool split_bf16) {
  const int tz = (d + W1_TZ - 1) / W1_TZ, ty = (h + W1_TY - 1) / W1_TY, tx = (w + W1_TX - 1) / W1_TX;
  const int nt = n * tz * ty * tx;
  int blocks = wgrad_c1_blocks(nt);
  MEDNET_REQUIRE(ws_bytes >= (size_t)blocks * 27 * cout * sizeof(float), MEDNET_E_WORKSPACE, "wgrad_c1: workspace too small");
  float* part = (float*)ws;
  const bool c1_mfma = dy_dtype == MEDNET_F16 ? mednet_f16::wgrad_c1_mfma_supported(cout, x_dtype, dy_dtype)
                                              : wgrad_c1_mfma_supported(cout, x_dtype, dy_dtype);
  if (split_bf16 && tuning_option("wgrad_c1_x3", 1) && wgrad_c1_x3_supported(cout, x_dtype, dy_dtype)) {  // fp32 storage mode
    blocks = wgrad_c1_x3_blocks(n, d, h, w);
    MEDNET_REQUIRE(ws_bytes >= (size_t)blocks * 27 * cout * sizeof(float), MEDNET_E_WORKSPACE, "wgrad_c1: workspace too small");
    const int rc = launch_wgrad_c1_x3(x, dy, part, n, d, h, w, s);
    if (rc) return rc;
    const size_t count = (size_t)27 * cout;
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, s, part, dw, count, blocks);
    return check_launch("wgrad_c1_reduce");
  }
  if (tuning_option("wgrad_c1_mfma", 1) && c1_mfma) {  // matrix-core form (16-bit modes)
    blocks = wgrad_c1_mfma_blocks(n, d, h, w);
    int rc = dy_dtype == MEDNET_F16 ? mednet_f16::launch_wgrad_c1_mfma(x, dy, part, n, d, h, w, cout, s, x_dtype)
                                    : launch_wgrad_c1_mfma(x, dy, part, n, d, h, w, cout, s, x_dtype);
    if (rc) return rc;
    const size_t count = (size_t)27 * cout;
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, s, part, dw, count, blocks);
    return check_launch("wgrad_c1_reduce");
  }
#define W1_GO(TX__, TDY__, CO__)                                                                                       \
  hipLaunchKernelGGL((wgrad_c1_kernel<TX__, TDY__, CO__>), dim3(blocks), dim3(256), 0, s, (const TX__*)x, (const TDY__*)dy, \
                     part, n, d, h, w, tz, ty, tx, nt)
#define W1_CO(TX__, TDY__)                         \
  do {                                             \
    if (cout == 8) W1_GO(TX__, TDY__, 8);          \
    else if (cout == 16) W1_GO(TX__, TDY__, 16);   \
    else if (cout == 32) W1_GO(TX__, TDY__, 32);   \
    else W1_GO(TX__, TDY__, 64);                   \
  } while (0)
  if (x_dtype == MEDNET_F32 && dy_dtype == MEDNET_F32) W1_CO(float, float);
  else if (x_dtype == MEDNET_F32 && dy_dtype == MEDNET_BF16) W1_CO(float, bf16);
  else if (x_dtype == MEDNET_BF16 && dy_dtype == MEDNET_BF16) W1_CO(bf16, bf16);
  else if (x_dtype == MEDNET_BF16 && dy_dtype == MEDNET_F32) W1_CO(bf16, float);
  else if (x_dtype == MEDNET_F32 && dy_dtype == MEDNET_F16) W1_CO(float, f16);
  else if (x_dtype == MEDNET_F16 && dy_dtype == MEDNET_F16) W1_CO(f16, f16);
  else W1_CO(f16, float);
#undef W1_CO
#undef W1_GO
  int rc = check_launch("wgrad_c1");
  if (rc) return rc;
  const size_t count = (size_t)27 * cout;
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, s, part, dw, count, blocks);
  return check_launch("wgrad_c1_reduce");
}

// ---- weight packing -------------------------------------------------------------------------------------------
// Pf[t][ci][co] and Pb[t][co][ci]; for a Conv3d source the Pb taps are mirrored (data gradient = correlation with
// the flipped kernel); for a ConvTranspose3d source (Cin,Cout,k,k,k) both keep the tap index.
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ Pf,
                                                           float* __restrict__ Pb, int cin, int cout, int T,
                                                           int transposed_src) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)cin * cout * T;
  if (i >= total) return;
  int t, ci, co;
  if (!transposed_src) {  // w[co][ci][t]
    t = (int)(i % T);
    ci = (int)((i / T) % cin);
    co = (int)(i / ((size_t)T * cin));
    Pf[((size_t)t * cin + ci) * cout + co] = w[i];
    Pb[((size_t)(T - 1 - t) * cout + co) * cin + ci] = w[i];
  } else {  // w[ci][co][t]
    t = (int)(i % T);
    co = (int)((i / T) % cout);
    ci = (int)(i / ((size_t)T * cout));
    Pf[((size_t)t * cin + ci) * cout + co] = w[i];
    Pb[((size_t)t * cout + co) * cin + ci] = w[i];
  }
}

int launch_pack_f32(const float* w, float* Pf, float* Pb, int cin, int cout, int T, int transposed_src,
                    hipStream_t s) {
  const size_t total = (size_t)cin * cout * T;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, Pf, Pb, cin, cout,
                     T, transposed_src);
  return check_launch("pack_weights");
}

}  // namespace mednet
