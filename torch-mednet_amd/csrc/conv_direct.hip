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
#pragma unroll
    for (int j = 0; j < COB; ++j) st(y, o + j, acc[j] + (skip ? ld(skip, o + j) : 0.f));
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
  return launch_direct_cob<bf16, float, MAP>(x, P, bias, skip, y, g, s);
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
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += part[(size_t)c * count + i];
  out[i] = s;
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
  else MEDNET_WG(bf16, float);
#undef MEDNET_WG
  int rc = check_launch("wgrad_direct");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, part, dw, count,
                     (int)chunks);
  return check_launch("wgrad_reduce");
}

// ---- per-channel sum over voxels (bias gradients) -----------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ x, float* __restrict__ out, int n,
                                                          size_t spatial, int c, int planar) {
  __shared__ float scratch[4];
  const int ch = blockIdx.x;
  float s = 0.f;
  const size_t total = (size_t)n * spatial;
  for (size_t i = threadIdx.x; i < total; i += 256) {
    const size_t nn = i / spatial, sp = i - nn * spatial;
    s += planar ? ld(x, (nn * c + ch) * spatial + sp) : ld(x, i * c + ch);
  }
  s = block_sum<4>(s, scratch);
  if (threadIdx.x == 0) out[ch] = s;
}

int launch_channel_sum(const void* x, float* out, int n, size_t spatial, int c, int planar, int dtype,
                       hipStream_t s) {
  if (dtype == MEDNET_F32)
    hipLaunchKernelGGL(channel_sum_kernel<float>, dim3(c), dim3(256), 0, s, (const float*)x, out, n, spatial, c, planar);
  else
    hipLaunchKernelGGL(channel_sum_kernel<bf16>, dim3(c), dim3(256), 0, s, (const bf16*)x, out, n, spatial, c, planar);
  return check_launch("channel_sum");
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
