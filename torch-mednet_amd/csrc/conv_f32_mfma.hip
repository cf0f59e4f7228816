// fp32 matrix-core kernels for the 3x3x3 convolution family: the 1e-3 PARITY mode (fp32 storage, config.precision "fp32")
// on v_mfma_f32_32x32x2_f32 instead of the VALU direct kernels (conv_direct.hip; those stay as the shape-generic fallback).
// The instruction is an exact fp32 fmaf chain (one rounding per product, fp32 accumulate: MI355X_MICROARCH.md, Matrix cores),
// so numerics are those of the direct kernels up to summation order; it runs at the fp32 vector peak (157 TFLOP/s, 1/16 of
// the bf16 rate), which makes everything around it cheap: plain synchronous LDS staging, two or three workgroups per CU
// to cover it, stores straight from the accumulators.
//
//   conv_f32_mfma_kernel<1>   nn.Conv3d 3^3 forward and data gradient (components.py:8-9,44), any Cin / Cout
//   conv_f32_mfma_kernel<2>   nn.ConvTranspose3d(k3,s2,p1,op1) data gradient (in = 2*out - 1 + tap)
//   convt_f32_mfma_kernel     nn.ConvTranspose3d forward + bias + skip (components.py:259-264,283-284), output-parity classes
//   wgrad_f32_mfma_kernel<S>  weight gradients of both (contraction over voxels: k = 2 voxels per MFMA)
//
// Operand maps of v_mfma_f32_32x32x2_f32 (one fp32 per lane and operand): lane l holds A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31]; D as for every 32x32 MFMA (row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), col = l & 31).
// Forward-type kernels: D[co][voxel], K = 8 input channels per LDS chunk = 4 MFMAs; a lane's 16-byte LDS piece holds the
// channels 4h .. 4h+3 of the chunk, MFMA j takes element j of both pieces (k = 0: channel j, k = 1: channel 4 + j).
// Weights are read from the fp32 tap-major images of the packed buffer (Pf[t][k][m] / Pb, the ones the direct kernels use).
#include "conv.h"

namespace mednet {

typedef __attribute__((ext_vector_type(4))) float f4;

template <int STRIDE>
struct F32Tile;
template <>
struct F32Tile<1> {
  static constexpr int TZ = 4, TY = 8, TX = 16;
};
template <>
struct F32Tile<2> {
  static constexpr int TZ = 2, TY = 4, TX = 16;
};

struct F32Args {
  const float* x;     // N x (id,ih,iw) x K, channels last
  const float* P;     // [27][K][M]
  const float* bias;  // nullable, M
  float* y;           // N x (od,oh,ow) x M
  int n, od, oh, ow, id, ih, iw, k, m;
  int tiles_z, tiles_y, tiles_x, ntiles, nkc, ncb;
};

// 16-byte piece = channels c0 .. c0+3 of one voxel row (zeros past K or outside the volume)
__device__ __forceinline__ f4 load_piece(const float* row, int c0, int K, bool in_vol) {
  f4 v = {0.f, 0.f, 0.f, 0.f};
  if (in_vol) {
    if ((K & 3) == 0) {
      if (c0 < K) v = *reinterpret_cast<const f4*>(row + c0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + j < K) v[j] = row[c0 + j];
    }
  }
  return v;
}

// weight slice of one chunk: w_lds[tap][h][co 32][4] <- P[tap][kc*8 + 4h + j][cb*32 + co]; thread `tid` moves elements
// it * 256 + tid.  Split into fetch (global -> registers, in flight during the previous chunk's MFMAs) and commit.
__device__ __forceinline__ void fetch_weights(const float* P, float (&w_reg)[27], int kc, int cb, int K, int M, int tid) {
#pragma unroll
  for (int it = 0; it < 27; ++it) {
    const int idx = it * 256 + tid;
    const int co = idx & 31, i = (idx >> 5) & 7, t = idx >> 8;
    const int ci = kc * 8 + i, m = cb * 32 + co;
    w_reg[it] = (ci < K && m < M) ? P[((size_t)t * K + ci) * M + m] : 0.f;
  }
}
__device__ __forceinline__ void commit_weights(float* w_lds, const float (&w_reg)[27], int tid) {
#pragma unroll
  for (int it = 0; it < 27; ++it) {
    const int idx = it * 256 + tid;
    const int co = idx & 31, i = (idx >> 5) & 7, t = idx >> 8;
    w_lds[((t * 2 + (i >> 2)) * 32 + co) * 4 + (i & 3)] = w_reg[it];
  }
}

template <int STRIDE>
__global__ __launch_bounds__(256, 2) void conv_f32_mfma_kernel(F32Args a) {
  using G = F32Tile<STRIDE>;
  constexpr int TZ = G::TZ, TY = G::TY, TX = G::TX;
  constexpr int HZ = STRIDE * (TZ - 1) + 3, HY = STRIDE * (TY - 1) + 3, HX = STRIDE * (TX - 1) + 3;
  constexpr int NV = HZ * HY * HX;
  constexpr int NTW = TZ * TY * TX / 32 / 4;
  constexpr int IN_ROUNDS = (2 * NV + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4* in_lds = reinterpret_cast<f4*>(smem);  // [2 k-halves][NV]
  f4* w_lds = in_lds + 2 * NV;               // [27][2][32]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const int tile = blockIdx.x / a.ncb, cb = blockIdx.x % a.ncb;
  int tt = tile;
  const int tx0 = (tt % a.tiles_x) * TX;
  tt /= a.tiles_x;
  const int ty0 = (tt % a.tiles_y) * TY;
  tt /= a.tiles_y;
  const int tz0 = (tt % a.tiles_z) * TZ;
  const int n = tt / a.tiles_z;
  const float* xs = a.x + (size_t)n * a.id * a.ih * a.iw * a.k;

  // staging plan of this thread: halo piece p = it * 256 + tid -> (voxel p >> 1, k-half p & 1)
  long long goff[IN_ROUNDS];  // element offset of the voxel row, -1 outside the volume, -2 past the halo
#pragma unroll
  for (int it = 0; it < IN_ROUNDS; ++it) {
    const int v = (it * 256 + tid) >> 1;
    long long off = -2;
    if (v < NV) {
      const int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
      const int gz = STRIDE * tz0 - 1 + hz, gy = STRIDE * ty0 - 1 + hy, gx = STRIDE * tx0 - 1 + hx;
      off = (gz >= 0 && gz < a.id && gy >= 0 && gy < a.ih && gx >= 0 && gx < a.iw)
                ? (((long long)gz * a.ih + gy) * a.iw + gx) * a.k
                : -1;
    }
    goff[it] = off;
  }
  int lbase[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int g = wv * NTW + t;
    const int lz = g / (TY / 2), ly = (g % (TY / 2)) * 2 + (r >> 4), lx = r & 15;
    lbase[t] = ((STRIDE * lz) * HY + STRIDE * ly) * HX + STRIDE * lx + h * NV;
  }

  f32x16 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = cb * 32 + 8 * q + 4 * h + j;
        acc[t][q * 4 + j] = (a.bias && co < a.m) ? a.bias[co] : 0.f;
      }

  // staging is software-pipelined through registers: chunk kc + 1 is in flight while chunk kc is on the matrix cores
  f4 in_reg[IN_ROUNDS];
  float w_reg[27];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) {
      const int p = it * 256 + tid;
      in_reg[it] = load_piece(xs + (goff[it] >= 0 ? goff[it] : 0), kc * 8 + (p & 1) * 4, a.k, goff[it] >= 0);
    }
    fetch_weights(a.P, w_reg, kc, cb, a.k, a.m, tid);
  };
  fetch(0);
  for (int kc = 0; kc < a.nkc; ++kc) {
    __syncthreads();  // the previous chunk's operand reads are done
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) {
      const int p = it * 256 + tid;
      if (goff[it] != -2) in_lds[(p & 1) * NV + (p >> 1)] = in_reg[it];
    }
    commit_weights(reinterpret_cast<float*>(w_lds), w_reg, tid);
    __syncthreads();
    if (kc + 1 < a.nkc) fetch(kc + 1);
#pragma unroll 3
    for (int tap = 0; tap < 27; ++tap) {
      const int toff = ((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3;
      const f4 wa = w_lds[(tap * 2 + h) * 32 + r];
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const f4 xb = in_lds[lbase[t] + toff];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], xb[j], acc[t], 0, 0, 0);
      }
    }
  }

  // stores straight from the accumulators: a lane holds four 16-byte pieces (co = 8q + 4h .. +3) of its voxel's row
  const size_t ovol = (size_t)a.od * a.oh * a.ow;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int g = wv * NTW + t;
    const int oz = tz0 + g / (TY / 2), oy = ty0 + (g % (TY / 2)) * 2 + (r >> 4), ox = tx0 + (r & 15);
    if (oz < a.od && oy < a.oh && ox < a.ow) {
      float* yp = a.y + ((size_t)n * ovol + ((size_t)oz * a.oh + oy) * a.ow + ox) * a.m;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co0 = cb * 32 + 8 * q + 4 * h;
        if ((a.m & 3) == 0) {
          if (co0 < a.m) {
            const f4 o = {acc[t][q * 4], acc[t][q * 4 + 1], acc[t][q * 4 + 2], acc[t][q * 4 + 3]};
            *reinterpret_cast<f4*>(yp + co0) = o;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co0 + j < a.m) yp[co0 + j] = acc[t][q * 4 + j];
        }
      }
    }
  }
}

template <int STRIDE>
static int launch_f32(const void* x, const float* P, const float* bias, void* y, int n, int od, int oh, int ow, int id, int ih,
                      int iw, int k, int m, hipStream_t s) {
  using G = F32Tile<STRIDE>;
  constexpr int HZ = STRIDE * (G::TZ - 1) + 3, HY = STRIDE * (G::TY - 1) + 3, HX = STRIDE * (G::TX - 1) + 3;
  constexpr size_t lds = ((size_t)2 * HZ * HY * HX + 27 * 2 * 32) * 16;
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  F32Args a;
  a.x = (const float*)x; a.P = P; a.bias = bias; a.y = (float*)y;
  a.n = n; a.od = od; a.oh = oh; a.ow = ow; a.id = id; a.ih = ih; a.iw = iw; a.k = k; a.m = m;
  a.tiles_z = (od + G::TZ - 1) / G::TZ;
  a.tiles_y = (oh + G::TY - 1) / G::TY;
  a.tiles_x = (ow + G::TX - 1) / G::TX;
  a.ntiles = n * a.tiles_z * a.tiles_y * a.tiles_x;
  a.nkc = (k + 7) / 8;
  a.ncb = (m + 31) / 32;
  MEDNET_REQUIRE((double)a.ntiles * a.ncb < 2147483647.0, MEDNET_E_UNSUPPORTED, "conv_f32_mfma: grid too large");
  static bool attr_set[3] = {false, false, false};
  if (!attr_set[STRIDE]) {
    if (hipFuncSetAttribute((const void*)conv_f32_mfma_kernel<STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(MEDNET_E_HIP, "conv_f32_mfma: cannot raise dynamic LDS to %zu", lds);
    attr_set[STRIDE] = true;
  }
  hipLaunchKernelGGL((conv_f32_mfma_kernel<STRIDE>), dim3((unsigned)(a.ntiles * a.ncb)), dim3(256), lds, s, a);
  return check_launch("conv_f32_mfma");
}

bool conv_f32_mfma_enabled() { return tuning_option("f32_mfma", 1) != 0; }

int launch_conv_f32_mfma(const void* x, const float* P, const float* bias, void* y, int n, int d, int h, int w, int k, int m,
                         hipStream_t s) {
  return launch_f32<1>(x, P, bias, y, n, d, h, w, d, h, w, k, m, s);
}
int launch_convt_dgrad_f32_mfma(const void* dy, const float* Pb, void* dx, int n, int d, int h, int w, int cin, int cout,
                                hipStream_t s) {
  // dx (d,h,w; Cin) <- dy (2d,2h,2w; Cout): contraction over Cout
  return launch_f32<2>(dy, Pb, nullptr, dx, n, d, h, w, 2 * d, 2 * h, 2 * w, cout, cin, s);
}

// ================================================================================================== ConvTranspose3d forward
// out[2j + p] = bias + skip + sum over the taps k of parity class p of W[k] * x[j + delta_k]   (per dim: k=1 -> p=0,d=0;
// k=0 -> p=1,d=1; k=2 -> p=1,d=0).  A workgroup owns a 2x4x16 brick of INPUT voxels (4x8x32 outputs), a wave one N-tile of
// 32 input voxels with all 8 output parity classes in registers, as in the bf16 kernel (convt_fwd_mfma_kernel).
struct CtF32Args {
  const float* x;
  const float* P;  // Pf[27][Cin][Cout]
  const float* bias;
  const float* skip;
  float* y;
  int n, id, ih, iw, k, m;
  int tiles_z, tiles_y, tiles_x, ntiles, nkc, ncb;
};

__global__ __launch_bounds__(256, 2) void convt_f32_mfma_kernel(CtF32Args a) {
  constexpr int TZ = 2, TY = 4, TX = 16, HZ = TZ + 1, HY = TY + 1, HX = TX + 1, NV = HZ * HY * HX;
  constexpr int IN_ROUNDS = (2 * NV + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4* in_lds = reinterpret_cast<f4*>(smem);
  f4* w_lds = in_lds + 2 * NV;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const int tile = blockIdx.x / a.ncb, cb = blockIdx.x % a.ncb;
  int tt = tile;
  const int tx0 = (tt % a.tiles_x) * TX;
  tt /= a.tiles_x;
  const int ty0 = (tt % a.tiles_y) * TY;
  tt /= a.tiles_y;
  const int tz0 = (tt % a.tiles_z) * TZ;
  const int n = tt / a.tiles_z;
  const float* xs = a.x + (size_t)n * a.id * a.ih * a.iw * a.k;
  long long goff[IN_ROUNDS];
#pragma unroll
  for (int it = 0; it < IN_ROUNDS; ++it) {
    const int v = (it * 256 + tid) >> 1;
    long long off = -2;
    if (v < NV) {
      const int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
      const int gz = tz0 + hz, gy = ty0 + hy, gx = tx0 + hx;
      off = (gz < a.id && gy < a.ih && gx < a.iw) ? (((long long)gz * a.ih + gy) * a.iw + gx) * a.k : -1;
    }
    goff[it] = off;
  }
  const int lz = wv / (TY / 2), ly = (wv % (TY / 2)) * 2 + (r >> 4), lx = r & 15;
  const int lbase = (lz * HY + ly) * HX + lx + h * NV;
  f32x16 acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;

  f4 in_reg[IN_ROUNDS];
  float w_reg[27];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) {
      const int p = it * 256 + tid;
      in_reg[it] = load_piece(xs + (goff[it] >= 0 ? goff[it] : 0), kc * 8 + (p & 1) * 4, a.k, goff[it] >= 0);
    }
    fetch_weights(a.P, w_reg, kc, cb, a.k, a.m, tid);
  };
  fetch(0);
  for (int kc = 0; kc < a.nkc; ++kc) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IN_ROUNDS; ++it) {
      const int p = it * 256 + tid;
      if (goff[it] != -2) in_lds[(p & 1) * NV + (p >> 1)] = in_reg[it];
    }
    commit_weights(reinterpret_cast<float*>(w_lds), w_reg, tid);
    __syncthreads();
    if (kc + 1 < a.nkc) fetch(kc + 1);
    f4 xb[8];
#pragma unroll
    for (int dl = 0; dl < 8; ++dl) xb[dl] = in_lds[lbase + (((dl >> 2) & 1) * HY + ((dl >> 1) & 1)) * HX + (dl & 1)];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
      const int pc = (kz != 1) * 4 + (ky != 1) * 2 + (kx != 1);  // output parity class of this tap
      const int dl = (kz == 0) * 4 + (ky == 0) * 2 + (kx == 0);  // input offset of this tap
      const f4 wa = w_lds[(tap * 2 + h) * 32 + r];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[pc] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], xb[dl][j], acc[pc], 0, 0, 0);
    }
  }
  const int od = 2 * a.id, oh = 2 * a.ih, ow = 2 * a.iw;
  const int jz = tz0 + lz, jy = ty0 + ly, jx = tx0 + lx;
  if (jz < a.id && jy < a.ih && jx < a.iw) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int oz = 2 * jz + (p >> 2), oy = 2 * jy + ((p >> 1) & 1), ox = 2 * jx + (p & 1);
      const size_t o = ((((size_t)n * od + oz) * oh + oy) * ow + ox) * a.m;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = cb * 32 + 8 * q + 4 * h + j;
          if (co < a.m) a.y[o + co] = acc[p][q * 4 + j] + (a.bias ? a.bias[co] : 0.f) + (a.skip ? a.skip[o + co] : 0.f);
        }
    }
  }
}

int launch_convt_fwd_f32_mfma(const void* x, const float* Pf, const float* bias, const void* skip, void* y, int n, int d, int h,
                              int w, int cin, int cout, hipStream_t s) {
  constexpr size_t lds = ((size_t)2 * 3 * 5 * 17 + 27 * 2 * 32) * 16;
  CtF32Args a;
  a.x = (const float*)x; a.P = Pf; a.bias = bias; a.skip = (const float*)skip; a.y = (float*)y;
  a.n = n; a.id = d; a.ih = h; a.iw = w; a.k = cin; a.m = cout;
  a.tiles_z = (d + 1) / 2; a.tiles_y = (h + 3) / 4; a.tiles_x = (w + 15) / 16;
  a.ntiles = n * a.tiles_z * a.tiles_y * a.tiles_x;
  a.nkc = (cin + 7) / 8;
  a.ncb = (cout + 31) / 32;
  hipLaunchKernelGGL(convt_f32_mfma_kernel, dim3((unsigned)(a.ntiles * a.ncb)), dim3(256), lds, s, a);
  return check_launch("convt_f32_mfma");
}

// ================================================================================================== weight gradients
//   R[tap][a][b] = sum_v A[v][a] * B[map(v, tap)][b]      conv:  A = dy, B = x,  map = v + tap - 1;   dw[(a*KB + b)*27 + tap]
//                                                         convT: A = x,  B = dy, map = 2v - 1 + tap
// k = 2 x-consecutive voxels per MFMA: lane (a = l & 31, k = l >> 5) reads A_lds[v0 + k][a], lane (b, k) reads
// B_lds[map(v0 + k, tap)][b]; with [voxel][32 channel] fp32 rows a half-wave reads 128 contiguous bytes (no bank conflict).
// A wave owns 7 of the 27 taps (7 x 16 accumulators), a workgroup a 32x32 channel-block pair and every `splits`-th brick;
// per-workgroup slabs are summed in a fixed order by wgrad_f32_reduce_kernel (no atomics).
template <int STRIDE>
struct WgF32Tile;
template <>
struct WgF32Tile<1> {
  static constexpr int TZ = 2, TY = 4, TX = 16;
};
template <>
struct WgF32Tile<2> {
  static constexpr int TZ = 1, TY = 4, TX = 8;
};

struct WgF32Args {
  const float* A;
  const float* B;
  float* part;  // [workgroup][27][32][32]
  int n, ad, ah, aw, bd, bh, bw, ka, kb;
  int tiles_z, tiles_y, tiles_x, ntiles, nab, nbb, splits;
};

template <int STRIDE>
__global__ __launch_bounds__(256, 2) void wgrad_f32_mfma_kernel(WgF32Args a) {
  using G = WgF32Tile<STRIDE>;
  constexpr int TZ = G::TZ, TY = G::TY, TX = G::TX;
  constexpr int HZ = STRIDE * (TZ - 1) + 3, HY = STRIDE * (TY - 1) + 3, HX = STRIDE * (TX - 1) + 3;
  constexpr int NA = TZ * TY * TX, NB = HZ * HY * HX;
  constexpr int KSTEPS = NA / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* A_lds = reinterpret_cast<float*>(smem);  // [NA][32]
  float* B_lds = A_lds + NA * 32;                 // [NB][32]
  const int pair = blockIdx.x / a.splits, split = blockIdx.x % a.splits;
  const int ab = pair / a.nbb, bb = pair % a.nbb;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, hk = lane >> 5;
  f32x16 acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  int toff[7];  // the wave with 6 taps recomputes tap 26 in its 7th slot (discarded at write-out)
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = wv + 4 * i < 27 ? wv + 4 * i : 26;
    toff[i] = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * 32;
  }
  for (int tile = split; tile < a.ntiles; tile += a.splits) {
    int tt = tile;
    const int tx0 = (tt % a.tiles_x) * TX;
    tt /= a.tiles_x;
    const int ty0 = (tt % a.tiles_y) * TY;
    tt /= a.tiles_y;
    const int tz0 = (tt % a.tiles_z) * TZ;
    const int n = tt / a.tiles_z;
    __syncthreads();  // the previous brick is consumed
    for (int c = tid; c < NA * 8; c += 256) {
      const int v = c >> 3, part = c & 7;
      const int gz = tz0 + v / (TX * TY), gy = ty0 + (v / TX) % TY, gx = tx0 + v % TX;
      const bool in_vol = gz < a.ad && gy < a.ah && gx < a.aw;
      const float* row = a.A + ((((size_t)n * a.ad + (in_vol ? gz : 0)) * a.ah + (in_vol ? gy : 0)) * a.aw + (in_vol ? gx : 0)) * a.ka;
      *reinterpret_cast<f4*>(A_lds + v * 32 + part * 4) = load_piece(row, ab * 32 + part * 4, a.ka, in_vol);
    }
    for (int c = tid; c < NB * 8; c += 256) {
      const int v = c >> 3, part = c & 7;
      const int gz = STRIDE * tz0 - 1 + v / (HX * HY), gy = STRIDE * ty0 - 1 + (v / HX) % HY, gx = STRIDE * tx0 - 1 + v % HX;
      const bool in_vol = gz >= 0 && gz < a.bd && gy >= 0 && gy < a.bh && gx >= 0 && gx < a.bw;
      const float* row = a.B + ((((size_t)n * a.bd + (in_vol ? gz : 0)) * a.bh + (in_vol ? gy : 0)) * a.bw + (in_vol ? gx : 0)) * a.kb;
      *reinterpret_cast<f4*>(B_lds + v * 32 + part * 4) = load_piece(row, bb * 32 + part * 4, a.kb, in_vol);
    }
    __syncthreads();
#pragma unroll 4
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int v = 2 * ks + hk;  // this lane's voxel of the k-step (x-fastest brick order; TX is even)
      const int lx = v % TX, ly = (v / TX) % TY, lz = v / (TX * TY);
      const float fa = A_lds[v * 32 + r];
      const float* brow = B_lds + (((STRIDE * lz) * HY + STRIDE * ly) * HX + STRIDE * lx) * 32 + r;
#pragma unroll
      for (int i = 0; i < 7; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, brow[toff[i]], acc[i], 0, 0, 0);
    }
  }
  float* out = a.part + (size_t)blockIdx.x * 27 * 1024;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = wv + 4 * i;
    if (tap < 27) {
#pragma unroll
      for (int j = 0; j < 16; ++j) out[((size_t)tap * 32 + (j & 3) + 8 * (j >> 2) + 4 * hk) * 32 + r] = acc[i][j];
    }
  }
}

// dw[(a*KB + b)*27 + tap] = sum over the splits of part[(pair*splits + split)][tap][a%32][b%32], fixed order
__global__ __launch_bounds__(256) void wgrad_f32_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int ka,
                                                               int kb, int nbb, int splits) {
  const size_t total = (size_t)((ka + 31) / 32) * nbb * 1024 * 27;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int b32 = (int)(e % 32), a32 = (int)((e / 32) % 32), tap = (int)((e / 1024) % 27);
  const int pair = (int)(e / (1024 * 27));
  const int ab = pair / nbb, bb = pair % nbb;
  const float* src = part + ((size_t)pair * splits) * 27 * 1024 + (size_t)tap * 1024 + a32 * 32 + b32;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 4 <= splits; k += 4) {
    s0 += src[(size_t)k * 27 * 1024];
    s1 += src[(size_t)(k + 1) * 27 * 1024];
    s2 += src[(size_t)(k + 2) * 27 * 1024];
    s3 += src[(size_t)(k + 3) * 27 * 1024];
  }
  for (; k < splits; ++k) s0 += src[(size_t)k * 27 * 1024];
  if (ab * 32 + a32 < ka && bb * 32 + b32 < kb) dw[((size_t)(ab * 32 + a32) * kb + bb * 32 + b32) * 27 + tap] = (s0 + s1) + (s2 + s3);
}

template <int STRIDE>
static void wgf32_plan(int n, int ad, int ah, int aw, int ka, int kb, WgF32Args& a) {
  using G = WgF32Tile<STRIDE>;
  a.tiles_z = (ad + G::TZ - 1) / G::TZ;
  a.tiles_y = (ah + G::TY - 1) / G::TY;
  a.tiles_x = (aw + G::TX - 1) / G::TX;
  a.ntiles = n * a.tiles_z * a.tiles_y * a.tiles_x;
  a.nab = (ka + 31) / 32;
  a.nbb = (kb + 31) / 32;
  const int pairs = a.nab * a.nbb;
  int splits = (512 + pairs - 1) / pairs;  // about two workgroups per CU
  if (splits > a.ntiles) splits = a.ntiles;
  if (splits < 1) splits = 1;
  a.splits = splits;
}

size_t wgrad_f32_mfma_ws_bytes(int n, int d, int h, int w, int ka, int kb, int stride2) {
  WgF32Args a;
  if (stride2) wgf32_plan<2>(n, d, h, w, ka, kb, a);
  else wgf32_plan<1>(n, d, h, w, ka, kb, a);
  return (size_t)a.nab * a.nbb * a.splits * 27 * 1024 * sizeof(float);
}

template <int STRIDE>
static int launch_wgf32(const void* A, const void* B, float* dw, int n, int ad, int ah, int aw, int bd, int bh, int bw, int ka,
                        int kb, void* ws, size_t ws_bytes, hipStream_t s) {
  using G = WgF32Tile<STRIDE>;
  constexpr int HZ = STRIDE * (G::TZ - 1) + 3, HY = STRIDE * (G::TY - 1) + 3, HX = STRIDE * (G::TX - 1) + 3;
  constexpr size_t lds = ((size_t)G::TZ * G::TY * G::TX + (size_t)HZ * HY * HX) * 128;
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  WgF32Args a;
  a.A = (const float*)A; a.B = (const float*)B; a.part = (float*)ws;
  a.n = n; a.ad = ad; a.ah = ah; a.aw = aw; a.bd = bd; a.bh = bh; a.bw = bw; a.ka = ka; a.kb = kb;
  wgf32_plan<STRIDE>(n, ad, ah, aw, ka, kb, a);
  const size_t need = (size_t)a.nab * a.nbb * a.splits * 27 * 1024 * sizeof(float);
  MEDNET_REQUIRE(ws_bytes >= need, MEDNET_E_WORKSPACE, "wgrad_f32_mfma: workspace %zu < %zu", ws_bytes, need);
  static bool attr_set[3] = {false, false, false};
  if (!attr_set[STRIDE]) {
    if (hipFuncSetAttribute((const void*)wgrad_f32_mfma_kernel<STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(MEDNET_E_HIP, "wgrad_f32_mfma: cannot raise dynamic LDS to %zu", lds);
    attr_set[STRIDE] = true;
  }
  hipLaunchKernelGGL((wgrad_f32_mfma_kernel<STRIDE>), dim3(a.nab * a.nbb * a.splits), dim3(256), lds, s, a);
  int rc = check_launch("wgrad_f32_mfma");
  if (rc) return rc;
  const size_t total = (size_t)a.nab * a.nbb * 1024 * 27;
  hipLaunchKernelGGL(wgrad_f32_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.part, dw, ka, kb, a.nbb, a.splits);
  return check_launch("wgrad_f32_reduce");
}

int launch_wgrad_f32_mfma(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cin, int cout, void* ws,
                          size_t ws_bytes, hipStream_t s) {
  return launch_wgf32<1>(dy, x, dw, n, d, h, w, d, h, w, cout, cin, ws, ws_bytes, s);  // A = dy (Cout rows), B = x (Cin cols)
}
int launch_convt_wgrad_f32_mfma(const void* x, const void* dy, float* dw, int n, int d, int h, int w, int cin, int cout, void* ws,
                                size_t ws_bytes, hipStream_t s) {
  return launch_wgf32<2>(x, dy, dw, n, d, h, w, 2 * d, 2 * h, 2 * w, cin, cout, ws, ws_bytes, s);  // A = x, B = dy at 2v - 1 + tap
}

}  // namespace mednet
